"""Sliding-window temporal inference (BASELINE config 5), the caller of the hot path in
scripts/run_action_classification_temporal_inf.py:74-130 + scripts/module_wrapper.py (VideoActionClassifier,
ActionProposalFromVideoTemporalDataset), minus video decoding: the input is an already decoded uint8 frame stream on
the GPU.  Window list, frame sampling, resize and normalisation follow the reference exactly; the per-window work
(gather + resize + normalise, then the MViT forward) runs in HIP kernels, batched, and windows are sharded over ranks.
"""
import pickle

import numpy as np
import torch

from .. import _hip
from .. import distributed as du


def get_proposals(num_frames, prop_length=64, prop_stride=16):
    """(t0, t1) per window; tail windows overrun the stream (module_wrapper.py:246-253)."""
    return [(i, i + prop_length) for i in range(0, num_frames, prop_stride)]


def frame_idxs_uniform(t0, t1, frame_length, video_num_frame):
    """torch.linspace(t0, t1, frame_length).clamp(0, n-1).long() (module_wrapper.py:384-397)."""
    index = torch.linspace(t0, t1, frame_length)
    return torch.clamp(index, 0, video_num_frame - 1).long()


class SlidingWindowClassifier(object):
    """VideoActionClassifier.inference (module_wrapper.py:403-611) over a decoded stream.

    model: an eval-mode MViT on the GPU (build_model); frame_length/frame_stride/proposal_* as the reference CLI
    (run_action_classification_temporal_inf.py:17-72; asserts proposal_length == frame_length * frame_stride, :76).
    """

    def __init__(self, model, frame_length=16, frame_stride=4, proposal_length=64, proposal_stride=16, frame_size=448,
                 batch_size=8, mean=0.45, std=0.225):
        assert proposal_length == frame_length * frame_stride
        self.model = model.eval()
        self.frame_length, self.proposal_length, self.proposal_stride = frame_length, proposal_length, proposal_stride
        self.frame_size, self.batch_size, self.mean, self.std = frame_size, batch_size, mean, std

    def window_frame_indices(self, windows, num_frames, device):
        """int32 [len(windows), T] on the device: the frames each window samples (module_wrapper.py:304-370)."""
        idx = torch.stack([frame_idxs_uniform(t0, t1, self.frame_length, num_frames) for t0, t1 in windows]).to(torch.int32)
        return idx.to(device)

    def preprocess(self, frames_u8, windows, idx=None, out=None):
        """frames_u8: uint8 [N,H,W,3] on the GPU; windows: list of (t0,t1) -> fp32 [len(windows),3,T,S,S].
        idx: the windows' rows of window_frame_indices, already on the device (run_views() uploads the whole table once: an upload
        per batch is a stream-ordered blocking copy, i.e. the host would wait for the previous batch's forward before it could
        enqueue the next one).  out: rows of a batch buffer to fill instead of a fresh tensor (a batch of run_views() may hold
        windows of two views: one launch per view into its rows)."""
        N, H, W, C = frames_u8.shape
        assert frames_u8.dtype == torch.uint8 and C == 3 and frames_u8.is_cuda and frames_u8.is_contiguous()
        idx = self.window_frame_indices(windows, N, frames_u8.device) if idx is None else idx
        S = self.frame_size
        if out is None:
            out = torch.empty(len(windows), 3, self.frame_length, S, S, dtype=torch.float32, device=frames_u8.device)
        assert out.is_contiguous() and tuple(out.shape) == (len(windows), 3, self.frame_length, S, S) and out.dtype == torch.float32
        assert idx.is_contiguous() and tuple(idx.shape) == (len(windows), self.frame_length) and idx.dtype == torch.int32
        _hip.check(_hip.lib().mvit_window_preprocess(_hip.ptr(frames_u8), _hip.ptr(idx), _hip.ptr(out), H, W, S, len(windows),
                                                     self.frame_length, self.mean, self.std,
                                                     torch.cuda.current_stream().cuda_stream), "window_preprocess")
        return out

    def batch_bounds(self, n):
        """[(i0, i1)] over n windows: batches of batch_size as the reference's DataLoader makes them (module_wrapper.py:384-397),
        except that a ragged tail of at most batch_size // 4 windows rides with the batch before it (57 windows at batch 8 ->
        6 x 8 + 9 instead of 7 x 8 + 1: a one-clip forward costs a third of an eight-clip one).  A window's scores do not depend on
        the batch it is in (every kernel of the forward is row- / clip-local)."""
        bs = self.batch_size
        cuts = list(range(0, n, bs)) + [n]
        if len(cuts) > 2 and 0 < cuts[-1] - cuts[-2] <= bs // 4:
            del cuts[-2]
        return list(zip(cuts[:-1], cuts[1:]))

    def pair_batches(self, n):
        """[(i0, i1)] over the n (view, window) pairs of one rank: round(n / batch_size) batches whose sizes differ by at most one
        (171 pairs at batch 8 -> 3 x 9 + 18 x 8; the 22 pairs of a rank at 8 GPUs -> 8 + 7 + 7), one more batch if that would put
        more than batch_size + batch_size // 4 clips into one -- no short last batch, whatever n is."""
        bs = self.batch_size
        if n <= 0:
            return []
        nb = max(1, (2 * n + bs) // (2 * bs))
        if -(-n // nb) > bs + bs // 4:
            nb += 1
        base, extra = divmod(n, nb)
        cuts = [0]
        for b in range(nb):
            cuts.append(cuts[-1] + base + (1 if b < extra else 0))
        return list(zip(cuts[:-1], cuts[1:]))

    @torch.no_grad()
    def run_views(self, views, shard=True):
        """BASELINE configs[4] as SURVEY section 8(e) shards it: the (view, window) pairs of ALL views (3 camera views x 57 windows =
        171) form one list, split rank-strided over the ranks (padded by wrapping: 176 at 8 ranks, not 3 x 64), every rank runs its
        pairs in batches that may cross a view boundary, ONE all_gather of the [n, 18] scores, ONE host copy.  Returns one result
        list per view, each exactly what the reference writes for that video: (t0, t1, float32[num_classes]) sorted by t0
        (run_action_classification_temporal_inf.py:99-130 runs the views one after the other on one GPU).  views: uint8
        [N_v, H, W, 3] tensors on the GPU (lengths may differ)."""
        assert len(views) > 0
        dev = views[0].device
        wins = [get_proposals(int(v.shape[0]), self.proposal_length, self.proposal_stride) for v in views]
        pairs = [(vi, wi) for vi, ws in enumerate(wins) for wi in range(len(ws))]
        if not pairs:
            return [[] for _ in views]
        world = du.get_world_size() if shard else 1
        mine = du.shard_indices(len(pairs), pad=True) if world > 1 else list(range(len(pairs)))
        # one index table for all of this rank's pairs, uploaded once
        idx_all = torch.stack([frame_idxs_uniform(*wins[pairs[j][0]][pairs[j][1]], self.frame_length, int(views[pairs[j][0]].shape[0]))
                               for j in mine]).to(torch.int32).to(dev)
        S = self.frame_size
        probs = []
        # (Measured and not kept, round 6: the front end of batch i + 1 enqueued under the forward of batch i on the least-loaded sub-batch
        # stream -- 706.1 / 706.5 against 709.4 / 704.4 clips/s for the serial order, profiles/r6_window_prefetch_ab.txt: the front end is
        # ~1 % of a batch; what separates this path from the bare forward's rate is the drain at the one host copy per call.)
        for i0, i1 in self.pair_batches(len(mine)):
            clips = torch.empty(i1 - i0, 3, self.frame_length, S, S, dtype=torch.float32, device=dev)
            a = i0
            while a < i1:                       # runs of one view inside the batch: one gather + resize launch each
                vi = pairs[mine[a]][0]
                b = a
                while b < i1 and pairs[mine[b]][0] == vi:
                    b += 1
                self.preprocess(views[vi], [wins[vi][pairs[mine[j]][1]] for j in range(a, b)], idx_all[a:b], out=clips[a - i0:b - i0])
                a = b
            probs.append(self.model([clips]).float())
        probs = torch.cat(probs, 0)
        ids = list(mine)
        if world > 1:
            probs = du.all_gather_cat(probs)          # the one data-path collective; the pair ids are a function of (n, world) alone
            ids = [j for r in range(world) for j in du.shard_indices(len(pairs), rank=r, world=world, pad=True)]
        probs = probs.cpu().numpy()                   # the one host copy (synchronises)
        core = self.model.module if hasattr(self.model, "module") else self.model
        if hasattr(core, "check_finite"):
            core.check_finite()                # (the copy above has synchronised: no extra wait) fp16 overflow under HIP.PRECISION auto raises here
        if not np.isfinite(probs).all():       # whatever the arithmetic: the scores are on the host now, a non-finite one never leaves this call
            raise FloatingPointError("SlidingWindowClassifier: non-finite scores for %d of %d windows (HIP.PRECISION %s); pin HIP.PRECISION bf16 "
                                     "or fp32 for this checkpoint" % (int((~np.isfinite(probs).all(1)).sum()), len(probs),
                                                                      getattr(getattr(core.cfg, "HIP", None), "PRECISION", "auto")))
        row = {}
        for k, j in enumerate(ids):            # padding duplicates: the first occurrence is kept (all copies are equal)
            row.setdefault(j, k)
        out = [[] for _ in views]
        for j, (vi, wi) in enumerate(pairs):
            t0, t1 = wins[vi][wi]
            out[vi].append((t0, t1, probs[row[j]].astype(np.float32)))
        for r in out:
            r.sort(key=lambda x: x[0])
        return out

    def run(self, frames_u8, shard=True):
        """One view: the reference's per-video result, list of (t0, t1, float32[num_classes]) sorted by t0
        (run_action_classification_temporal_inf.py:111-125).  With shard=True and an initialised process group the windows are
        split rank-strided over the ranks and gathered back (every rank returns the full list)."""
        return self.run_views([frames_u8], shard=shard)[0]

    @staticmethod
    def save(result, path):
        """The on-disk wire format consumed by the post-processing: pickle of the list
        (run_action_classification_temporal_inf.py:128-130)."""
        with open(path, "wb") as f:
            pickle.dump(result, f)
