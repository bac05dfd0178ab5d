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
