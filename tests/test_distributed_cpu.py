"""CPU, world_size 2, gloo: the multi-process helpers of the data-parallel path."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from aicity_action_amd import distributed as du


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    loss, top1, top5 = torch.tensor(1.0 + rank), torch.tensor(10.0 * (rank + 1)), torch.tensor([3.0, 4.0]) * (rank + 1)
    r = du.all_reduce([loss, top1, top5])
    part = torch.arange(3, dtype=torch.float32).view(3, 1) + 10 * rank
    g = du.all_gather_cat(part)
    shard = du.shard_indices(7)
    q.put((rank, [x.tolist() for x in r], g.flatten().tolist(), shard))
    dist.destroy_process_group()


def test_allreduce_gather_and_sharding_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, red, gathered, shard in res:
        assert red[0] == pytest.approx(1.5) and red[1] == pytest.approx(15.0) and red[2] == pytest.approx([4.5, 6.0])
        assert gathered == [0.0, 1.0, 2.0, 10.0, 11.0, 12.0]
    assert res[0][3] == [0, 2, 4, 6] and res[1][3] == [1, 3, 5, 0]      # padded by wrapping: equal counts per rank
    assert du.shard_indices(7, 1, 2, pad=False) == [1, 3, 5]
    assert [du.shard_indices(3, r, 8) for r in range(8)] == [[0], [1], [2], [0], [1], [2], [0], [1]]      # fewer items than ranks: wraps as often as needed
    assert du.shard_indices(0, 0, 2) == []
    assert du.get_world_size() == 1 and du.get_rank() == 0


class _ToyNet(torch.nn.Module):
    """CPU stand-in with the model's calling convention (list of one [B,3,T,H,W] tensor -> [B,classes])."""

    def __init__(self):
        super().__init__()
        self.fc = torch.nn.Linear(3, 6)

    def forward(self, x):
        y = self.fc(x[0].mean(dim=(2, 3, 4)))
        return y if self.training else torch.softmax(y, 1)


class _ToyOpt(object):
    def __init__(self, m):
        self.o = torch.optim.SGD(m.parameters(), lr=0.1)

    def set_lr(self, lr):
        for g in self.o.param_groups:
            g["lr"] = lr

    def zero_grad(self):
        self.o.zero_grad()

    def step(self):
        self.o.step()

    def state_dict(self):
        return self.o.state_dict()

    def load_state_dict(self, sd):
        self.o.load_state_dict(sd)


def _loop_worker(rank, world, port, q):
    import json
    import logging
    from aicity_action_amd import engine
    from aicity_action_amd.config import get_cfg
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = get_cfg()
    cfg.MODEL.LOSS_FUNC, cfg.MODEL.NUM_CLASSES, cfg.LOG_PERIOD = "cross_entropy", 6, 1
    cfg.SOLVER.MAX_EPOCH, cfg.SOLVER.BASE_LR, cfg.SOLVER.COSINE_END_LR, cfg.SOLVER.LR_POLICY = 1, 0.1, 0.0, "cosine"
    torch.manual_seed(0)
    model = _ToyNet()                      # identical weights on both ranks
    g = torch.Generator().manual_seed(10 + rank)
    loader = [([torch.randn(4, 3, 2, 4, 4, generator=g)], torch.randint(0, 6, (4,), generator=g), torch.arange(4), {})]
    lines = []

    class _H(logging.Handler):
        def emit(self, rec):
            lines.append(rec.getMessage())
    logging.getLogger("aicity_action_amd.engine").addHandler(_H())
    logging.getLogger("aicity_action_amd.engine").setLevel(logging.INFO)
    with torch.no_grad():
        local = torch.nn.functional.cross_entropy(model.train()(loader[0][0]), loader[0][1]).item()
    engine.train_epoch(loader, model, _ToyOpt(model), None, engine.TrainMeter(1, cfg), 0, cfg)
    it = [json.loads(l.split("json_stats: ")[1]) for l in lines if "train_iter" in l]
    # multi-view test: rank r holds clips {r, r+2} of 2 videos x 2 clips; predictions are gathered to every rank
    tm = engine.TestMeter(num_videos=2, num_clips=2, num_cls=6, overall_iters=1)
    tl = [([torch.randn(2, 3, 2, 4, 4, generator=g)], torch.tensor([1, 2]), torch.tensor([rank, rank + 2]), {})]
    st = engine.perform_test(tl, model, tm, cfg)
    q.put((rank, local, it[0]["loss"] if it else None, len(it), tm.clip_count.tolist(), st["split"]))
    dist.destroy_process_group()


def test_train_and_test_loops_world2():
    """engine.train_epoch averages loss / errors over the ranks in one collective and logs on rank 0 only; perform_test gathers
    every rank's clips into the view-sum ensemble (tools/train_net.py:284-287, tools/test_net.py:119-122)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_loop_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    mean_loss = 0.5 * (res[0][1] + res[1][1])
    assert res[0][3] == 1 and res[1][3] == 0                               # json_stats only from the master process
    assert res[0][2] == pytest.approx(mean_loss, abs=1e-4)                 # all-reduced (averaged) loss of the global batch
    for r in res:
        assert r[4] == [2, 2] and r[5] == "test_final"                     # both ranks saw all 4 clips


class _ToySlider(object):
    """SlidingWindowClassifier with the GPU front end replaced by a CPU gather (frame means): the sharding / batching / gather logic of
    run_views() is what runs here."""

    @staticmethod
    def make(batch_size):
        from aicity_action_amd.inference import SlidingWindowClassifier

        class S(SlidingWindowClassifier):
            def preprocess(self, frames_u8, windows, idx=None, out=None):
                idx = self.window_frame_indices(windows, frames_u8.shape[0], frames_u8.device) if idx is None else idx
                fr = frames_u8.float().mean(dim=(1, 2))[idx.long()]                     # [n, T, 3]
                val = fr.permute(0, 2, 1)[:, :, :, None, None].expand(-1, -1, -1, self.frame_size, self.frame_size)
                if out is None:
                    return val.contiguous()
                out.copy_(val)
                return out
        torch.manual_seed(0)
        return S(_ToyNet().eval(), frame_size=4, batch_size=batch_size)


def _toy_views():
    g = torch.Generator().manual_seed(3)
    return [torch.randint(0, 256, (n, 2, 2, 3), dtype=torch.uint8, generator=g) for n in (900, 900, 610)]


def _views_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = _ToySlider.make(8).run_views(_toy_views())
    q.put((rank, [[(a, b, p.tolist()) for a, b, p in r] for r in res]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_view_window_pairs_sharded_over_two_ranks_equal_the_per_view_runs(world):
    """BASELINE configs[4] as SURVEY 8(e) shards it: the (view, window) pairs of all views rank-strided over the ranks, ONE all_gather;
    every rank gets every view's full list, bit-equal to running each view alone in one process (the reference's serial order,
    run_action_classification_temporal_inf.py:99-130)."""
    from aicity_action_amd.inference.sliding_window import get_proposals
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_views_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=300) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    views = _toy_views()
    swc = _ToySlider.make(8)
    ref = [swc.run(v, shard=False) for v in views]
    one = swc.run_views(views, shard=False)
    for res in [one] + [g_[1] for g_ in got]:          # every rank holds every view's full list (8 ranks: 167 pairs in 168 padded slots)
        assert len(res) == 3
        for v, rr, r in zip(views, ref, res):
            assert [(a, b) for a, b, _ in r] == get_proposals(v.shape[0], 64, 16) == [(a, b) for a, b, _ in rr]
            for (_, _, p), (_, _, pr) in zip(r, rr):
                assert (torch.tensor(p, dtype=torch.float32) == torch.from_numpy(pr)).all()


def test_pair_batches_are_balanced_and_cover_every_pair_once():
    swc = _ToySlider.make(8)
    assert [b - a for a, b in swc.pair_batches(171)] == [9] * 3 + [8] * 18          # 3 views x 57 windows on one GPU
    assert [b - a for a, b in swc.pair_batches(22)] == [8, 7, 7]                    # 176 padded pairs / 8 ranks
    assert len(du.shard_indices(171, 0, 8, pad=True)) == 22                          # 176 slots, not 3 x 64 = 192
    for bs in (1, 3, 8, 16):
        s = _ToySlider.make(bs)
        for n in range(0, 200):
            bb = s.pair_batches(n)
            assert [a for a, _ in bb] == [0] * (n > 0) + [b for _, b in bb[:-1]] and (not bb or bb[-1][1] == n)
            sizes = [b - a for a, b in bb]
            assert not sizes or (max(sizes) - min(sizes) <= 1 and max(sizes) <= bs + max(bs // 4, 1))
