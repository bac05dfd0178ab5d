#!/usr/bin/env python3
"""bench.py -- clips/sec of the MViTv2-B 16x4 @448 hot path on MI355X (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode train|fwd] [--batch 8] [--crop 448]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Both forms work for N > 1: started WITHOUT a launcher (no WORLD_SIZE in the environment) `bench.py --gpus N` starts the second form
itself as a child process -- before anything in this process has touched the GPU -- on 127.0.0.1 and a free port, lets the ranks
print (rank 0: the ONE JSON line) and exits with the launcher's return code (non-zero if any rank failed).  The reference starts its
ranks the same way: one spawned process per GPU from the entry script (slowfast/utils/misc.py:292-322, multiprocessing.py:9-68).

A "step" is one pass of the hot path over one batch of synthetic clips per GPU ([8,3,16,448,448] N(0,1),
generated on the device before the timed region; random-init weights from the seeded generator).
  --mode train (default; BASELINE.json's metric is fwd+bwd): forward (drop-path 0.4, head dropout 0.5) + soft-target CE
      + backward + global-norm clip 1.0 + AdamW, fp32 master weights; N>1 = DistributedDataParallel over RCCL
      (one gradient all-reduce of 35.3 M fp32 per step, bucketed, overlapped with the per-block backward).
  --mode fwd: eval forward only (BASELINE configs[1]); clips sharded over ranks, no data-path collective.
  --mode loop: the same train step driven by aicity_action_amd.engine.train_epoch (the reference's loop: per-iteration LR,
      top-k errors, stat all-reduce, meters, json_stats lines) over a synthetic in-memory loader -- what a user of
      tools/run_net.py gets; must stay within a few % of --mode train (no per-iteration host sync).
  --mode window: BASELINE configs[4]: sliding-window inference over a synthetic 30 s 540p stream x 3 camera views
      (57 windows of 16x4 frames per view, GPU gather + cv2-style resize to 448 + normalise + forward), the 171 (view, window)
      pairs sharded rank-strided over the ranks, one all_gather of the scores ("strong" scaling: the stream is fixed).
Per-GPU work is fixed as N grows ("weak" scaling); the timed region is bracketed by barrier +
torch.cuda.synchronize() and the MAX over ranks is reported (wall clock over exactly K steps: the driver's contract; the
median of per-step HIP-event intervals is given beside it as `ms_per_step_event_median`).  Rank 0 prints ONE JSON
line with `roofline` (dominant kernel = fused attention, timed live with HIP events on the launch stream),
`cpu_baseline` (the oracle -- CPU restatement of the reference's unfused op sequence -- on the host cores) and, in the default
train mode, `forward`: the eval-forward record of BASELINE configs[1] (bf16 and fp16 MFMA builds: clips/s, fraction of the
MFMA roofline -- the north star's 30 % target is on this number -- and the logit error against the reference's golden vector).
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
TRAFFIC_FWD = "r6_attn_fwd_hbm_traffic.json"   # profiles/: PMC HBM bytes per attention launch (tools/traffic.sh via tools/r6_profiles.sh)
TRAFFIC_BWD = "r6_attn_bwd_hbm_traffic.json"
TRAFFIC_TAIL = "r6_block_tail_hbm_traffic.json"  # ... per launch of the fused block tail at the stage-3 shape


def attention_flops(geoms, B):
    return [4.0 * B * g.heads * g.lq * g.lk * 96 for g in geoms]


def self_launch(n):
    """`bench.py --gpus N` without an external launcher: run torch.distributed.run with N ranks as a CHILD process (this process has
    not imported torch and never touches the GPU, so nothing is exec'ed over an initialised HIP runtime) and hand back its return
    code; the ranks inherit stdout / stderr, so rank 0's JSON line is this command's JSON line."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")                  # torch.distributed.run would set 1 (and warn); the CPU legs are rank-0, N = 1 only
    # --standalone: the launcher itself picks AND HOLDS the rendezvous port (c10d store on port 0), so no other process can take it
    # between choosing and binding (a bind-close-reuse of a "free" port raced on shared boxes); --local-addr pins MASTER_ADDR to
    # 127.0.0.1 (the container hostname may not resolve)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--standalone", "--local-addr", "127.0.0.1",
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", default="train", choices=["train", "fwd", "window", "loop"])
    ap.add_argument("--batch", type=int, default=8, help="clips per GPU per step")
    ap.add_argument("--crop", type=int, default=448, choices=[224, 448])
    ap.add_argument("--precision", default=None, choices=["bf16", "fp16", "fp32"],
                    help="default: HIP.PRECISION auto = bf16 for the train modes, fp16 (the arithmetic that meets the 1e-3 logit gate) for fwd / window")
    ap.add_argument("--streams", type=int, default=3, help="inference: sub-batches on separate HIP streams (cfg HIP.STREAMS)")
    ap.add_argument("--train-streams", type=int, default=1, help="training: sub-batches on separate HIP streams (cfg HIP.TRAIN_STREAMS)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-forward-record", action="store_true", help="train mode: skip the extra eval-forward timing")
    ap.add_argument("--graph", action="store_true", help="loop mode: HIP.GRAPH_STEP (the train step as one replayed hipGraph)")
    args = ap.parse_args()
    if args.precision is None:
        args.precision = "bf16" if args.mode in ("train", "loop") else "fp16"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

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
        raise SystemExit("--gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    ndev = torch.cuda.device_count()
    dev_index = local_rank if local_rank < ndev else 0      # a launcher that narrows each rank's visibility to one GPU
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend = os.environ.get("MVIT_BENCH_BACKEND", "nccl")      # "gloo": tests that run two ranks on ONE GPU (RCCL refuses that)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    yaml = "MVITV2_FULL_B_16x4_CONV_448.yaml" if args.crop == 448 else "MVITV2_FULL_B_16x4_CONV.yaml"
    cfg = load_config(os.path.join(ROOT, "configs", "Aicity", yaml), ["NUM_GPUS", 1, "HIP.PRECISION", args.precision, "HIP.STREAMS", args.streams,
                                                                      "HIP.TRAIN_STREAMS", args.train_streams])
    mv = copy.deepcopy(cfg.MVIT.to_dict())
    train = args.mode in ("train", "loop")
    manual_ddp = False
    if train and world > 1:
        if world <= ndev:
            cfg.NUM_GPUS = world          # build_model wraps DistributedDataParallel (slowfast/models/build.py:47-54)
        else:
            manual_ddp = True             # fewer visible devices than ranks: same wrap, applied here
    model = build_model(cfg, gpu_id=dev_index)
    if manual_ddp:
        from aicity_action_amd.models.build import wrap_ddp
        model = wrap_ddp(model, cfg, dev_index)
    core = model.module if hasattr(model, "module") else model
    load_synth_weights(core, 0)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    clip = torch.randn(args.batch, 3, 16, args.crop, args.crop, device=dev, generator=g)
    if args.mode == "window":
        from aicity_action_amd.inference import SlidingWindowClassifier
        model.eval()
        swc = SlidingWindowClassifier(model, frame_size=args.crop, batch_size=args.batch)
        gs = torch.Generator(device=dev).manual_seed(99)
        views = [torch.randint(0, 256, (900, 540, 960, 3), device=dev, dtype=torch.uint8, generator=gs) for _ in range(3)]
        n_windows = 3 * 57

        def step():
            res = swc.run_views(views)          # the 171 (view, window) pairs rank-strided over the ranks, one all_gather, one host copy
            return torch.from_numpy(res[0][0][2])
    elif args.mode == "loop":
        import logging
        from aicity_action_amd import engine
        from aicity_action_amd.solver import construct_optimizer
        logging.getLogger("aicity_action_amd.engine").setLevel(logging.WARNING)
        model.train()
        opt = construct_optimizer(model, cfg)
        labels = torch.arange(args.batch, device=dev) % cfg.MODEL.NUM_CLASSES
        cfg.LOG_PERIOD = 10                                             # configs/Aicity/*.yaml value
        cfg.HIP.GRAPH_STEP = bool(args.graph)

        class _Loader(list):
            pass

        def run_epoch(n, epoch):
            loader = _Loader([([clip], labels, torch.arange(args.batch), {})] * n)
            engine.train_epoch(loader, model, opt, None, engine.TrainMeter(n, cfg), epoch, cfg)

        step = None
    elif train:
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

    ev_ms = None
    if args.mode == "loop":
        run_epoch(max(args.warmup, 1), 0)
        barrier()
        t0 = time.perf_counter()
        run_epoch(args.steps, 1)
        barrier()
        dt = time.perf_counter() - t0
        out = torch.zeros(1, device=dev)
    else:
        for _ in range(args.warmup):
            out = step()
        barrier()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        t0 = time.perf_counter()
        evs[0].record()
        for i in range(args.steps):
            out = step()
            evs[i + 1].record()
        barrier()
        dt = time.perf_counter() - t0
        iv = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps))
        ev_ms = iv[len(iv) // 2]
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    per_rank = [{"rank": 0, "device": dev_index, "ms_per_step": round(dt / args.steps * 1e3, 4)}]
    allreduce_ok = None
    if world > 1:
        ones = torch.ones(8, device=dev)
        dist.all_reduce(ones)                      # every rank contributes: the sum must be the world size the launcher asked for
        allreduce_ok = bool((ones == float(world)).all().item())
        assert allreduce_ok, "all-reduce of ones gave %s on world size %d" % (ones.tolist(), world)
        mine = torch.tensor([dt, float(dev_index)], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [{"rank": i, "device": int(e[1].item()), "ms_per_step": round(e[0].item() / args.steps * 1e3, 4)} for i, e in enumerate(every)]
    for r_ in per_rank:                            # what each rank did on its own clock (the headline uses the slowest rank's time for all)
        r_["clips_per_s"] = round((n_windows / world if args.mode == "window" else args.batch) / (r_["ms_per_step"] * 1e-3), 2)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()
    assert torch.isfinite(out).all()
    if args.mode == "train":
        # the optimizer skips a step whose global gradient norm is not finite, so a broken backward would leave the loss finite --
        # and NaN operands switch so little that such a step even times FASTER (data-dependent clock): check the norm itself
        gn = opt.last_grad_norm
        assert gn is not None and bool(torch.isfinite(gn).all()), "non-finite gradient norm in the timed steps"
    clips_per_s = world * args.batch * args.steps / dt
    if args.mode == "window":
        clips_per_s = n_windows * args.steps / dt

    # ---- in-step cost of the dominant family: one EXTRA step (after the K timed ones) with two HIP events around every fused-attention
    # launch, recorded on the stream the launch goes to (the sub-batch streams of the inference forward included) ------------------------
    instep = None
    if not args.no_kernel_timing and args.mode in ("train", "fwd"):
        # EVERY rank runs the extra step (a DDP train step all-reduces its gradients: rank 0 alone would wait for the others forever);
        # only rank 0 brackets its launches with events
        if rank == 0:
            _hip.ATT_TIMER = []
        try:
            step()
            barrier()
            rec = _hip.ATT_TIMER or []
        finally:
            _hip.ATT_TIMER = None
    if rank == 0 and not args.no_kernel_timing and args.mode in ("train", "fwd"):
        instep = {}
        for kind in ("fwd", "bwd"):
            rows = [(fl, e0.elapsed_time(e1)) for k_, fl, e0, e1 in rec if k_ == kind]
            if rows:
                instep[kind] = {"launches": len(rows), "flops": sum(r_[0] for r_ in rows), "ms": sum(r_[1] for r_ in rows)}

    # ---- dominant kernel family (fused attention): live HIP-event timing on the launch stream -------------------
    # fwd mode : attn_fwd kernel, algorithmic FLOPs = sum_blocks 4*B*h*Lq*Lk*96.
    # train    : the attention backward (delta + dQ pass + dK/dV pass per block) is the largest item of the step;
    #            algorithmic FLOPs = 2x the forward's (train = 3x forward, recompute not credited).
    roofline = None
    extra_rooflines = {}
    if rank == 0 and not args.no_kernel_timing and args.mode != "window":
        L = core._lib()
        act = _hip.F32 if args.precision == "fp32" else _hip.BF16
        adt = torch.float32 if act == _hip.F32 else core._half_dtype()
        st = torch.cuda.current_stream().cuda_stream
        flops = attention_flops(core.geoms, args.batch)
        peak = PEAK_F32_TFLOPS if args.precision == "fp32" else PEAK_BF16_TFLOPS
        reps = 5

        def timed(fn):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / reps

        fwd_ms, bwd_ms, per_f, per_b = 0.0, 0.0, [], []
        # the launches of the timed region: inference runs HIP.STREAMS sub-batches, so each attention launch covers B / streams clips
        # (uneven splits -- 8 clips on 3 streams = 3, 3, 2 -- are timed on the larger launch; `achieved` is a rate, `launches` counts all)
        sub = min(args.streams, args.batch // 2) if not train else 1
        sub = sub if sub > 1 else 1
        clips_pl = -(-args.batch // sub)
        flops = [f * clips_pl / args.batch for f in flops]
        for gm, fl in zip(core.geoms, flops):
            B_, h_ = clips_pl, gm.heads
            q = torch.randn(B_, h_, gm.lq, 96, device=dev).to(adt)
            k = torch.randn(B_, h_, gm.lk, 96, device=dev).to(adt)
            v = torch.randn(B_, h_, gm.lk, 96, device=dev).to(adt)
            o = torch.empty(B_, gm.lq, h_ * 96, device=dev, dtype=adt)
            lse = torch.empty(B_, h_, gm.lq, device=dev)
            ms = timed(lambda: _hip.check(_hip.attention_fwd(L, q, k, v, o, lse, B_, h_, gm.lq, gm.lk, 96 ** -0.5, 1, act, st)))      # as the model calls it
            fwd_ms += ms
            per_f.append(round(fl / ms / 1e9, 1))
            if train:
                do = torch.randn(B_, gm.lq, h_ * 96, device=dev).to(adt)
                dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
                ws = torch.empty(L.mvit_attention_bwd_workspace_bytes(B_, h_, gm.lq, gm.lk) // 4, device=dev)
                ms = timed(lambda: _hip.check(L.mvit_attention_bwd(_hip.ptr(q), _hip.ptr(k), _hip.ptr(v), _hip.ptr(o), _hip.ptr(lse),
                                                                    _hip.ptr(do), _hip.ptr(dq), _hip.ptr(dk), _hip.ptr(dv), _hip.ptr(ws),
                                                                    B_, h_, gm.lq, gm.lk, 96 ** -0.5, 1, act, st)))
                bwd_ms += ms
                per_b.append(round(2 * fl / ms / 1e9, 1))
                del do, dq, dk, dv, ws
            del q, k, v, o, lse

        traffic_src = {}

        def pmc_traffic(fname, key, tag):
            """HBM bytes per launch from the committed PMC measurement of this exact workload (tools/traffic.sh: rocprofv3 --pmc
            FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 correction applied); None for any other configuration.  The file
            records the commit of the kernels it was measured on (`measured_at_commit`), echoed here so staleness is visible."""
            path = os.path.join(ROOT, "profiles", fname)
            if not (args.batch == 8 and args.crop == 448 and args.precision in ("bf16", "fp16") and os.path.exists(path)):      # (both 16-bit builds move the same bytes)
                return None
            try:
                doc = json.load(open(path))
                rec = doc["by_clips_per_launch"].get(str(clips_pl))
                traffic_src[tag] = {"file": "profiles/" + fname, "measured_at_commit": doc.get("measured_at_commit", "unrecorded")}
                return round(float(rec[key]), 0) if rec else None
            except Exception:
                return None

        def rl(name, tot_flops, tot_ms, per_block, traffic=None, inst=None):
            """`achieved` / `frac` = the IN-STEP figure when the extra instrumented step ran (HIP events around every launch of the family
            inside a real step: between the step's other launches, beside its side streams); `achieved_alone` / `frac_alone` = the same
            kernels on randn operands, back to back with nothing else on the GPU."""
            ach_alone = tot_flops / (tot_ms * 1e-3) / 1e12
            ach = ach_alone if inst is None else inst["flops"] / (inst["ms"] * 1e-3) / 1e12
            d = {"kernel": name,
                 "timed": ("in-step: two HIP events around every launch of this family during one extra step after the K timed ones, on the launch's own stream; "
                           "`*_alone`: randn operands, back to back on an otherwise idle GPU") if inst is not None else
                          "alone, randn operands, back to back on the launch stream",
                 "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                 "achieved_alone": round(ach_alone, 2), "frac_alone": round(ach_alone / peak, 4), "traffic": traffic,
                 "launches": len(flops) * sub, "clips_per_launch": clips_pl, "avg_launch_ms": round(tot_ms / len(flops), 4),
                 "algorithmic_gflop_per_launch_avg": round(tot_flops / len(flops) / 1e9, 2), "tflops_per_block_alone": per_block}
            if inst is not None:
                d["in_step_ms"] = round(inst["ms"], 4)
                d["in_step_launches"] = inst["launches"]
                d["alone_ms"] = round(tot_ms, 4)        # one launch per block at `clips_per_launch` clips (in-step: every launch of the step)
            return d
        sfx = args.precision if act else "f32"
        fwd_rl = rl("attn_fwd_w64_kernel (%s)" % sfx if act else "attn_fwd_f32_kernel", sum(flops), fwd_ms, per_f,
                    pmc_traffic(TRAFFIC_FWD, "traffic_bytes_per_launch", "attention_fwd"), (instep or {}).get("fwd"))
        if train:
            roofline = rl("mvit_attention_bwd (attn_bwd_delta + attn_bwd_dq + attn_bwd_dkv kernels, %s)" % sfx, 2 * sum(flops), bwd_ms, per_b,
                          pmc_traffic(TRAFFIC_BWD, "traffic_bytes_per_call", "attention_bwd"), (instep or {}).get("bwd"))
            extra_rooflines["roofline_attention_fwd"] = fwd_rl
        else:
            roofline = fwd_rl
            if act:
                # second family of the forward: the fused block tail (proj + norm2 + fc1 + GELU + fc2 + residual in one kernel, csrc/mlp_fused.hip),
                # timed alone at the steady stage-3 shape (11 of 16 blocks); algorithmic FLOPs = 2 M C^2 (proj) + 16 M C^2 (fc1 + fc2)
                g3 = core.geoms[len(core.geoms) // 2]
                Mt, Ct = clips_pl * g3.lq, g3.dim_out
                blk = core.blocks[len(core.geoms) // 2]
                tk = core._tail_packed(blk, act)
                if tk is not None:
                    o_ = torch.randn(Mt, Ct, device=dev).to(adt)
                    r_ = torch.randn(Mt, Ct, device=dev)
                    out_ = torch.empty_like(r_)
                    ms = timed(lambda: _hip.check(L.mvit_block_tail_fwd(_hip.ptr(o_), _hip.ptr(r_), _hip.ptr(tk), _hip.ptr(blk.mlp.fc2.bias), _hip.ptr(out_), Mt, Ct,
                                                                         4 * Ct, blk.norm2.eps, act, st)))
                    fl_t = 18.0 * Mt * Ct * Ct
                    ach = fl_t / (ms * 1e-3) / 1e12
                    extra_rooflines["roofline_block_tail"] = {
                        "kernel": "mlp_fused_kernel<12, 1, true> (mvit_block_tail_fwd, %s), M = %d x C = %d" % (sfx, Mt, Ct), "bound": "mfma", "achieved": round(ach, 2),
                        "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": pmc_traffic(TRAFFIC_TAIL, "traffic_bytes_per_launch", "block_tail"),
                        "launches": 11 * sub, "clips_per_launch": clips_pl, "avg_launch_ms": round(ms, 4), "algorithmic_gflop_per_launch": round(fl_t / 1e9, 2),
                        "algorithmic_hbm_bytes_per_launch": Mt * Ct * 10}
                    del o_, r_, out_
        roofline["traffic_source"] = traffic_src

    # ---- train mode: the eval-forward record of BASELINE configs[1] beside the headline (north star: >= 30 % on THIS number) ----
    forward_rec = None
    if args.mode == "train" and not args.no_forward_record and args.crop == 448:
        import numpy as np
        forward_rec = {}
        gold = np.load(os.path.join(ROOT, "tests", "golden", "mvit_full448.npz"))
        gmeta = json.loads(bytes(gold["meta"]).decode())
        from aicity_action_amd.utils.synth import synth_clip
        gclip = synth_clip(1, 16, 448, gmeta["clip_seed"]).to(dev)
        for prec in ("fp16", "bf16"):        # fp16 first: the arithmetic HIP.PRECISION "auto" runs inference in, and the one that meets the gate
            cfg_f = load_config(os.path.join(ROOT, "configs", "Aicity", yaml), ["NUM_GPUS", 1, "HIP.PRECISION", prec, "HIP.STREAMS", args.streams])
            torch.cuda.empty_cache()
            mf = build_model(cfg_f, gpu_id=dev_index).eval()
            load_synth_weights(mf, 0)
            with torch.no_grad():
                _, lg = mf._forward_hip(gclip, return_logits=True)
                err = float(np.abs(lg.float().cpu().numpy() - gold["logits"]).max())
                for _ in range(5):
                    mf([clip])
                wins = []
                for _w in range(2):     # two windows of 20 steps; the MEAN is the reported rate (the faster one beside it)
                    barrier()
                    t0 = time.perf_counter()
                    for _ in range(20):
                        mf([clip])
                    barrier()
                    wins.append(time.perf_counter() - t0)
                fdt = sum(wins) / len(wins)
            ft = torch.tensor([fdt], device=dev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(ft, op=dist.ReduceOp.MAX)
            fdt = ft.item()
            cps = world * args.batch * 20 / fdt
            forward_rec[prec] = {"clips_per_s": round(cps, 2), "ms_per_step": round(fdt / 20 * 1e3, 4), "steps": 20, "warmup": 5,
                                 "model_roofline_frac": round(cps / world * GFLOP_PER_CLIP[448] / 1e3 / PEAK_BF16_TFLOPS, 4),
                                 "windows_ms_per_step": [round(w_ / 20 * 1e3, 4) for w_ in wins], "protocol": "mean of 2 windows of 20 steps",
                                 "best_window_clips_per_s": round(world * args.batch * 20 / min(wins), 2),
                                 "logit_err_vs_golden": float("%.3g" % err), "gate_1e-3_met": bool(err <= 1e-3),
                                 "golden": "tests/golden/mvit_full448.npz (reference CPU fp32 logits, B=1; gate 1e-3)"}
            del mf
        forward_rec["workload"] = "MViTv2-B 16x4 crop=448 eval forward, synthetic clips, BS=%d per GPU, HIP.STREAMS %d (BASELINE configs[1])" % (
            args.batch, args.streams)
        forward_rec["target_frac"] = 0.30
        met = [p_ for p_ in ("fp16", "bf16") if forward_rec[p_]["gate_1e-3_met"]]
        forward_rec["headline_dtype"] = met[0] if met else None      # the fastest-listed arithmetic whose logits are within 1e-3 of the reference's
        forward_rec["headline"] = forward_rec[met[0]] if met else None

    # ---- train mode, N = 1: BASELINE configs[4] beside the headline: sliding-window inference over 3 synthetic 30 s 540p views -------
    window_rec = None
    if args.mode == "train" and not args.no_forward_record and args.crop == 448 and world == 1:
        from aicity_action_amd.inference import SlidingWindowClassifier
        cfg_w = load_config(os.path.join(ROOT, "configs", "Aicity", yaml), ["NUM_GPUS", 1, "HIP.STREAMS", args.streams])     # HIP.PRECISION auto -> fp16 in eval
        torch.cuda.empty_cache()
        mw = build_model(cfg_w, gpu_id=dev_index).eval()
        load_synth_weights(mw, 0)
        swc = SlidingWindowClassifier(mw, frame_size=448, batch_size=args.batch)
        gs = torch.Generator(device=dev).manual_seed(99)
        views = [torch.randint(0, 256, (900, 540, 960, 3), device=dev, dtype=torch.uint8, generator=gs) for _ in range(3)]
        with torch.no_grad():
            prec_w = mw.precision                                 # what run() (no_grad) computes in: HIP.PRECISION auto -> fp16
        res = swc.run_views(views)                              # warm-up pass (also the shape check below)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            res = swc.run_views(views)
        torch.cuda.synchronize()
        wdt = (time.perf_counter() - t0) / 2
        window_rec = {"workload": "sliding-window inference: 3 views x 900 frames 540x960 uint8 (synthetic), 57 windows / view of 16 frames (stride 4), "
                                  "GPU gather + 8-bit INTER_LINEAR resize to 448 + normalise + forward (BASELINE configs[4], one GPU)",
                      "windows": sum(len(r) for r in res), "last_t1": int(res[0][-1][1]), "clips_per_s": round(3 * 57 / wdt, 2),
                      "seconds_per_30s_stream_3_views": round(wdt, 4), "seconds_per_view": round(wdt / 3, 4), "precision": prec_w,
                      "parity_note": "cv2 absent in this image -- front-end parity unpinned (bit-exact against oracle/window_oracle.py only)"}
        del mw, swc, views
        torch.cuda.empty_cache()

    # ---- CPU baseline: the oracle on the host cores, bounded sample (BASELINE.md section 4 as far as ~60 s allow) --------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.mode != "window":      # N = 1 only (the other ranks would idle behind it)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import mvit_oracle as O
        from aicity_action_amd.utils.synth import synth_state_dict
        from aicity_action_amd.models.mvit import MViT as _M
        cores = min(os.cpu_count() or 1, 32)     # more threads than this slows the oracle down on a 256-core host
        cpu_model = "unknown"
        try:
            for ln in open("/proc/cpuinfo"):
                if ln.startswith("model name"):
                    cpu_model = ln.split(":", 1)[1].strip()
                    break
        except OSError:
            pass
        sd448 = {k: v.detach().cpu() for k, v in core.state_dict().items()} if args.crop == 448 else None
        cfg224 = load_config(os.path.join(ROOT, "configs", "Aicity", "MVITV2_FULL_B_16x4_CONV.yaml"), ["NUM_GPUS", 0])
        sd224 = synth_state_dict({k_: v_.shape for k_, v_ in _M(cfg224).state_dict().items()}, 0)
        mv224 = copy.deepcopy(cfg224.MVIT.to_dict())
        c224 = torch.randn(1, 3, 16, 224, 224)

        def fwd_time(sd_, c_, mv_, n, threads):
            torch.set_num_threads(threads)
            ts = []
            for _ in range(n):
                t0 = time.perf_counter()
                with torch.no_grad():
                    O.forward(sd_, c_, mv_)
                ts.append(time.perf_counter() - t0)
            return sorted(ts)[len(ts) // 2]
        fwd_time(sd224, c224, mv224, 1, cores)                                  # warm-up
        f224 = fwd_time(sd224, c224, mv224, 3, cores)                           # median of 3
        f224_1t = fwd_time(sd224, c224, mv224, 1, 1)
        f448 = fwd_time(sd448, clip[:1].cpu(), mv, 1, cores) if sd448 is not None else None
        torch.set_num_threads(cores)
        extra = {"forward_224_s_per_clip": round(f224, 4), "forward_224_1thread_s_per_clip": round(f224_1t, 4),
                 "forward_448_s_per_clip": None if f448 is None else round(f448, 4), "cpu_model": cpu_model,
                 "protocol": "1 warm-up + median of 3 @224, 1 timed @448, 1 timed 1-thread @224 (BASELINE.md section 4, bounded)"}
        if train:
            import psutil
            big = args.crop == 448 and psutil.virtual_memory().available >= 96 * 2 ** 30   # ~20 GB of fp32 score matrices per clip
            sd_t, c_t, mv_t, crop_cpu = (sd448, clip[:1].cpu(), mv, 448) if big else (sd224, c224, mv224, 224)
            mv_t = dict(mv_t, DROPPATH_RATE=0.0)
            y1 = torch.zeros(1, cfg.MODEL.NUM_CLASSES)
            y1[0, 0] = 1.0
            sdg = {k_: v_.clone().requires_grad_(True) for k_, v_ in sd_t.items()}
            t0 = time.perf_counter()
            out_c, _ = O.forward(sdg, c_t, mv_t, training=True)
            O.soft_target_cross_entropy(out_c, y1).backward()
            cdt = time.perf_counter() - t0
            cpu = {"value": round(1.0 / cdt, 4), "unit": "clips/s", "cores": cores, "kind": "port",
                   "sample": "1 x forward+backward (torch autograd over the oracle) B=1 @%d fp32 after the forward warm-ups "
                             "(oracle/mvit_oracle.py, torch CPU, %d threads)" % (crop_cpu, cores)}
        else:
            val = f448 if f448 is not None else f224
            cpu = {"value": round(1.0 / val, 4), "unit": "clips/s", "cores": cores, "kind": "port",
                   "sample": "forward B=1 @%d fp32 (oracle/mvit_oracle.py, torch CPU, %d threads)" % (args.crop, cores)}
        cpu.update(extra)

    if rank == 0:
        gf = GFLOP_PER_CLIP[args.crop] * (3.0 if train else 1.0)   # train step = 3x forward FLOPs
        peak = PEAK_F32_TFLOPS if args.precision == "fp32" else PEAK_BF16_TFLOPS   # fp16 MFMA rate = bf16 rate on gfx950
        line = {
            "metric": "clips/sec (node) MViTv2-B 16x4@%d %s" % (args.crop, args.mode),
            "value": round(clips_per_s, 3), "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "ms_per_step_event_median": None if ev_ms is None else round(ev_ms, 4),
            "timing": "wall clock over the K steps between barrier + synchronize (driver contract); event median = per-step hipEvent intervals",
            "higher_is_better": True,
            "scaling": "strong" if args.mode == "window" else "weak",
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
            "world_size": dist.get_world_size() if world > 1 else 1,      # as the process group saw it
            "allreduce_of_ones_ok": allreduce_ok,
            "backend": (dist.get_backend() + (" (RCCL)" if backend == "nccl" else "")) if world > 1 else None,
            "ranks": per_rank, "ms_per_step_rank_min": min(r_["ms_per_step"] for r_ in per_rank),
            "ms_per_step_rank_max": max(r_["ms_per_step"] for r_ in per_rank),
        }
        if args.mode == "window":
            line["config"]["workload"] = ("sliding-window inference: 3 views x 900 frames 540x960 uint8 (synthetic), 57 windows/view of "
                                          "16 frames (stride 4), resize to %d, bf16 forward, batch %d (BASELINE configs[4])" % (args.crop, args.batch))
            line["config"]["parallelism"] = "dp%d ((view, window) pairs sharded rank-strided: %d per rank, ONE all_gather of [n,18] scores)" % (world, -(-n_windows // world))
            line["seconds_per_30s_stream_3views"] = round(dt / args.steps, 4)
            line["parity_note"] = ("front end (gather + 8-bit INTER_LINEAR resize + normalise) is bit-exact vs oracle/window_oracle.py; "
                                   "cv2 is absent from this image, so that restatement of cv2.resize is itself unpinned (SURVEY 8f rank 1: parity unpinned)")
        if args.mode == "loop":
            line["config"]["workload"] += " driven by engine.train_epoch (reference loop order, LOG_PERIOD 10, no per-iteration host sync%s)" % (
                "; the step replayed as one hipGraph" if args.graph else "")
        if forward_rec is not None:
            line["forward"] = forward_rec
        if window_rec is not None:
            line["window"] = window_rec
        line.update(extra_rooflines)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
