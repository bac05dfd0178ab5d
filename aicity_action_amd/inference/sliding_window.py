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

    def preprocess(self, frames_u8, windows, idx=None):
        """frames_u8: uint8 [N,H,W,3] on the GPU; windows: list of (t0,t1) -> fp32 [len(windows),3,T,S,S].
        idx: the windows' rows of window_frame_indices, already on the device (run() uploads the whole view's table once: an upload
        per batch is a stream-ordered blocking copy, i.e. the host would wait for the previous batch's forward before it could
        enqueue the next one)."""
        N, H, W, C = frames_u8.shape
        assert frames_u8.dtype == torch.uint8 and C == 3 and frames_u8.is_cuda and frames_u8.is_contiguous()
        idx = self.window_frame_indices(windows, N, frames_u8.device) if idx is None else idx
        S = self.frame_size
        out = torch.empty(len(windows), 3, self.frame_length, S, S, dtype=torch.float32, device=frames_u8.device)
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

    @torch.no_grad()
    def run(self, frames_u8, shard=True):
        """Returns the reference's per-video result: list of (t0, t1, float32[num_classes]) sorted by t0
        (run_action_classification_temporal_inf.py:111-125).  With shard=True and an initialised process group the
        windows are split rank-strided over the ranks and gathered back (every rank returns the full list)."""
        N = frames_u8.shape[0]
        windows = get_proposals(N, self.proposal_length, self.proposal_stride)
        world = du.get_world_size() if shard else 1
        mine = du.shard_indices(len(windows), pad=True) if world > 1 else list(range(len(windows)))
        probs = []
        idx_all = self.window_frame_indices([windows[j] for j in mine], N, frames_u8.device)
        for i0, i1 in self.batch_bounds(len(mine)):
            chunk = [windows[j] for j in mine[i0:i1]]
            clips = self.preprocess(frames_u8, chunk, idx_all[i0:i1])
            probs.append(self.model([clips]).float())
        probs = torch.cat(probs, 0)
        if world > 1:
            ids = torch.tensor(mine, device=probs.device, dtype=torch.int64)
            probs, ids = du.all_gather_cat(probs), du.all_gather_cat(ids)
            order = torch.argsort(ids, stable=True)
            keep = torch.ones_like(order, dtype=torch.bool)
            sid = ids[order]
            keep[1:] = sid[1:] != sid[:-1]                 # drop the padding duplicates
            probs = probs[order][keep]
        probs = probs.cpu().numpy()
        core = self.model.module if hasattr(self.model, "module") else self.model
        if hasattr(core, "check_finite"):
            core.check_finite()                # (the copy above has synchronised: no extra wait) fp16 overflow under HIP.PRECISION auto raises here
        if not np.isfinite(probs).all():       # whatever the arithmetic: the scores are on the host now, a non-finite one never leaves this call
            raise FloatingPointError("SlidingWindowClassifier: non-finite scores for %d of %d windows (HIP.PRECISION %s); pin HIP.PRECISION bf16 "
                                     "or fp32 for this checkpoint" % (int((~np.isfinite(probs).all(1)).sum()), len(probs),
                                                                      getattr(getattr(core.cfg, "HIP", None), "PRECISION", "auto")))
        out = [(t0, t1, probs[k].astype(np.float32)) for k, (t0, t1) in enumerate(windows)]
        out.sort(key=lambda x: x[0])
        return out

    @staticmethod
    def save(result, path):
        """The on-disk wire format consumed by the post-processing: pickle of the list
        (run_action_classification_temporal_inf.py:128-130)."""
        with open(path, "wb") as f:
            pickle.dump(result, f)
