#!/usr/bin/env python3
"""bench.py -- clips/sec of the MViTv2-B 16x4 @448 hot path on MI355X (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode train|fwd] [--batch 8] [--crop 448]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic clips per GPU ([8,3,16,448,448] N(0,1),
generated on the device before the timed region; random-init weights from the seeded generator).
  --mode train (default; BASELINE.json's metric is fwd+bwd): forward (drop-path 0.4, head dropout 0.5) + soft-target CE
      + backward + global-norm clip 1.0 + AdamW, fp32 master weights; N>1 = DistributedDataParallel over RCCL
      (one gradient all-reduce of 35.3 M fp32 per step, bucketed, overlapped with the per-block backward).
  --mode fwd: eval forward only (BASELINE configs[1]); clips sharded over ranks, no data-path collective.
Per-GPU work is fixed as N grows ("weak" scaling); the timed region is bracketed by barrier +
torch.cuda.synchronize() and the MAX over ranks is reported.  Rank 0 prints ONE JSON
line with `roofline` (dominant kernel = fused attention, timed live with HIP events on the launch stream) and
`cpu_baseline` (the oracle -- CPU restatement of the reference's unfused op sequence -- on the host cores).
"""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_CLIP = {448: 856.45, 224: 127.73}   # SURVEY.md section 8d (2 FLOP/MAC, GEMM + conv terms)
PEAK_BF16_TFLOPS = 2500.0                      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3


def attention_flops(geoms, B):
    return [4.0 * B * g.heads * g.lq * g.lk * 96 for g in geoms]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", default="train", choices=["train", "fwd"])
    ap.add_argument("--batch", type=int, default=8, help="clips per GPU per step")
    ap.add_argument("--crop", type=int, default=448, choices=[224, 448])
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from aicity_action_amd import _hip
    from aicity_action_amd.config import load_config
    from aicity_action_amd.models import build_model
    from aicity_action_amd.utils.synth import load_synth_weights

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    yaml = "MVITV2_FULL_B_16x4_CONV_448.yaml" if args.crop == 448 else "MVITV2_FULL_B_16x4_CONV.yaml"
    cfg = load_config(os.path.join(ROOT, "configs", "Aicity", yaml), ["NUM_GPUS", 1, "HIP.PRECISION", args.precision])
    mv = copy.deepcopy(cfg.MVIT.to_dict())
    train = args.mode == "train"
    if train and world > 1:
        cfg.NUM_GPUS = world              # build_model wraps DistributedDataParallel (slowfast/models/build.py:47-54)
    model = build_model(cfg, gpu_id=local_rank)
    core = model.module if hasattr(model, "module") else model
    load_synth_weights(core, 0)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    clip = torch.randn(args.batch, 3, 16, args.crop, args.crop, device=dev, generator=g)
    if train:
        from aicity_action_amd.solver import construct_optimizer, get_lr_at_epoch, soft_target_cross_entropy
        model.train()
        opt = construct_optimizer(model, cfg)
        labels = torch.zeros(args.batch, cfg.MODEL.NUM_CLASSES, device=dev)
        labels[torch.arange(args.batch), torch.arange(args.batch) % cfg.MODEL.NUM_CLASSES] = 1.0
        it = [0]

        def step():
            opt.set_lr(get_lr_at_epoch(cfg, it[0] / 1000.0))       # LR is reset every iteration (train_net.py:113-115)
            it[0] += 1
            logits = model([clip])
            loss = soft_target_cross_entropy(logits, labels)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()                                              # fused grad-norm clip (1.0) + AdamW
            return loss.detach()
    else:
        model.eval()

        def step():
            with torch.no_grad():
                return model([clip])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()
    assert torch.isfinite(out).all()
    clips_per_s = world * args.batch * args.steps / dt

    # ---- dominant kernel (fused attention): live HIP-event timing on the launch stream --------------------
    roofline = None
    if rank == 0 and not args.no_kernel_timing:
        L = _hip.lib()
        act = _hip.BF16 if args.precision == "bf16" else _hip.F32
        adt = torch.bfloat16 if act == _hip.BF16 else torch.float32
        st = torch.cuda.current_stream().cuda_stream
        flops = attention_flops(core.geoms, args.batch)
        tot_ms = 0.0
        reps = 5
        per_block = []
        for gm, fl in zip(core.geoms, flops):
            q = torch.randn(args.batch, gm.heads, gm.lq, 96, device=dev).to(adt)
            k = torch.randn(args.batch, gm.heads, gm.lk, 96, device=dev).to(adt)
            v = torch.randn(args.batch, gm.heads, gm.lk, 96, device=dev).to(adt)
            o = torch.empty(args.batch, gm.lq, gm.heads * 96, device=dev, dtype=adt)

            def run():
                _hip.check(L.mvit_attention_fwd(_hip.ptr(q), _hip.ptr(k), _hip.ptr(v), _hip.ptr(o), None, args.batch, gm.heads,
                                                gm.lq, gm.lk, 96 ** -0.5, 1, act, st))
            run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1) / reps
            tot_ms += ms
            per_block.append(round(fl / ms / 1e9, 1))
        peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_F32_TFLOPS
        achieved = sum(flops) / (tot_ms * 1e-3) / 1e12
        roofline = {"kernel": "attn_fwd_%s_kernel" % ("bf16" if act else "f32"), "bound": "mfma",
                    "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                    "traffic": None, "launches": len(flops), "avg_launch_ms": round(tot_ms / len(flops), 4),
                    "algorithmic_gflop_per_launch_avg": round(sum(flops) / len(flops) / 1e9, 2),
                    "tflops_per_block": per_block}

    # ---- CPU baseline: the oracle on the host cores, bounded sample ------------------------------------------
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import mvit_oracle as O
        cores = min(os.cpu_count() or 1, 32)     # more threads than this slows the oracle down on a 256-core host
        torch.set_num_threads(cores)
        sd = {k: v.detach().cpu() for k, v in core.state_dict().items()}
        c1 = clip[:1].cpu()
        n_timed = 1 if args.crop == 448 else 8
        with torch.no_grad():
            O.forward(sd, c1, mv)                      # warm-up
            t0 = time.perf_counter()
            for _ in range(n_timed):
                O.forward(sd, c1, mv)
            cdt = time.perf_counter() - t0
        cpu = {"value": round(n_timed / cdt, 4), "unit": "clips/s", "cores": cores, "kind": "port",
               "sample": "%d x forward B=1 @%d fp32 (oracle/mvit_oracle.py, torch CPU, %d threads) after 1 warm-up"
                         % (n_timed, args.crop, cores)}

    if rank == 0:
        gf = GFLOP_PER_CLIP[args.crop] * (3.0 if train else 1.0)   # train step = 3x forward FLOPs
        peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_F32_TFLOPS
        line = {
            "metric": "clips/sec (node) MViTv2-B 16x4@%d %s" % (args.crop, args.mode),
            "value": round(clips_per_s, 3), "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": ("MViTv2-B 16x4 crop=%d %s train step (fwd+bwd+clip+AdamW), synthetic clips, BS=%d per GPU (BASELINE configs[2]/[3])"
                                    if train else
                                    "MViTv2-B 16x4 crop=%d %s forward-only, synthetic clips, BS=%d per GPU (BASELINE configs[1])")
                                   % (args.crop, args.precision, args.batch),
                       "global_batch": world * args.batch,
                       "parallelism": ("dp%d (DDP, RCCL gradient all-reduce)" % world) if train else ("dp%d (clips sharded, no collective)" % world),
                       "gflop_per_clip": gf},
            "model_roofline": {"bound": "mfma", "achieved": round(clips_per_s / world * gf / 1e3, 2), "peak": peak,
                               "unit": "TFLOP/s", "frac": round(clips_per_s / world * gf / 1e3 / peak, 4)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
