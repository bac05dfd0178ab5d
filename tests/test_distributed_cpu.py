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
