"""GPU: DistributedDataParallel over the hand-written backward.  Two processes (gloo backend, both on cuda:0 -- RCCL
refuses two ranks on one device) each take half of the batch through the same DDP wrap build_model applies
(slowfast/models/build.py:47-54); the averaged gradients must equal the single-process full-batch gradients."""
import os
import socket

import numpy as np
import pytest
import torch

from conftest import cfg_for_case, load_golden

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from aicity_action_amd.models import build_model
    from aicity_action_amd.solver import soft_target_cross_entropy
    from aicity_action_amd.utils.synth import load_synth_weights, synth_clip
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z, meta = load_golden("tiny_even")
    cfg = cfg_for_case(meta, "fp32", train=True)
    cfg.NUM_GPUS = 1
    model = build_model(cfg, gpu_id=0).train()
    load_synth_weights(model, 0)
    model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], output_device=0)
    clip = synth_clip(meta["batch"], meta["num_frames"], meta["crop"], meta["clip_seed"]).cuda()
    labels = torch.from_numpy(z["train.labels"]).cuda()
    n = meta["batch"] // world
    sl = slice(rank * n, (rank + 1) * n)
    loss = soft_target_cross_entropy(model([clip[sl]]), labels[sl])
    loss.backward()
    torch.cuda.synchronize()
    if rank == 0:
        q.put({k: p.grad.detach().cpu().numpy() for k, p in model.module.named_parameters()})
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_two_ranks_average_equals_full_batch():
    import torch.multiprocessing as mp
    from aicity_action_amd.models import build_model
    from aicity_action_amd.solver import soft_target_cross_entropy
    from aicity_action_amd.utils.synth import load_synth_weights, synth_clip
    z, meta = load_golden("tiny_even")
    assert meta["batch"] == 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    ddp_grads = q.get(timeout=300)
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    cfg = cfg_for_case(meta, "fp32", train=True)
    cfg.NUM_GPUS = 1
    model = build_model(cfg).train()
    load_synth_weights(model, 0)
    clip = synth_clip(meta["batch"], meta["num_frames"], meta["crop"], meta["clip_seed"]).cuda()
    labels = torch.from_numpy(z["train.labels"]).cuda()
    soft_target_cross_entropy(model([clip]), labels).backward()
    worst = 0.0
    for k, p in model.named_parameters():
        ref = p.grad.cpu().numpy()
        err = np.abs(ddp_grads[k] - ref).max() / max(1e-4, np.abs(ref).max())
        worst = max(worst, err)
    print("DDP(2 ranks) vs full batch: worst relative grad error %.2e" % worst)
    assert worst <= 1e-3
