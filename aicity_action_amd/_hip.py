"""ctypes binding of libmvit_hip.so (the C-ABI in include/mvit_hip.h).

The product path has NO fallback: if the library is missing or a kernel returns an error the
caller gets an exception.  ``stream`` arguments are raw hipStream_t handles
(``torch.cuda.current_stream().cuda_stream``).
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MVIT_HIP_LIB") or os.path.join(_HERE, "lib", "libmvit_hip.so")   # override: tools/ ablation builds only
LIB_PATH_F16 = os.environ.get("MVIT_HIP_LIB_F16") or os.path.join(_HERE, "lib", "libmvit_hip_f16.so")   # same sources, 16-bit activation type = IEEE half
CSRC = os.path.join(_HERE, "csrc")

F32, BF16 = 0, 1
EPI_BIAS, EPI_GELU, EPI_RESIDUAL = 1, 2, 4

_lib = None
_lib_f16 = None

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_l = ctypes.c_int64
c_f = ctypes.c_float

_SIGS = {
    "mvit_version": (ctypes.c_char_p, []),
    "mvit_strerror": (ctypes.c_char_p, [c_i]),
    "mvit_layernorm_fwd": (c_i, [c_p, c_p, c_p, c_p, c_l, c_i, c_f, c_i, c_p]),
    "mvit_linear_fwd": (c_i, [c_p, c_i, c_l, c_p, c_p, c_p, c_l, c_p, c_l, c_p, c_i, c_l, c_l, c_i, c_i, c_i, c_i, c_p]),
    "mvit_mlp_fused_pack_bytes": (c_l, [c_i, c_i]),
    "mvit_mlp_fused_pack": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p]),
    "mvit_mlp_fused_fwd": (c_i, [c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_f, c_i, c_p]),
    "mvit_block_tail_pack_bytes": (c_l, [c_i, c_i]),
    "mvit_block_tail_pack": (c_i, [c_p] * 8 + [c_i, c_i, c_p]),
    "mvit_block_tail_fwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_f, c_i, c_p]),
    "mvit_pool_conv_ln_fwd": (c_i, [c_p, c_l, c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_p]),
    "mvit_attention_fwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_i, c_i, c_p]),
    "mvit_maxpool_skip_fwd": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "mvit_stem_fwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "mvit_head_workspace_bytes": (c_l, [c_i, c_i, c_i]),
    "mvit_head_fwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p]),
    "mvit_cast_f32_to_bf16": (c_i, [c_p, c_p, c_l, c_p]),
    "mvit_cast_rows_f32_to_bf16": (c_i, [c_p, c_p, c_l, c_i, c_p, c_l, c_p]),
    "mvit_cast_transpose_f32_to_bf16": (c_i, [c_p, c_p, c_p, c_i, c_i, c_p]),
    "mvit_side_stream": (c_p, []),
    "mvit_side_fork": (c_i, [c_p]),
    "mvit_side_join": (c_i, [c_p]),
    "mvit_cast_desc_bytes": (c_i, []),
    "mvit_cast_transpose_multi": (c_i, [c_p, c_i, c_i, c_p]),
    "mvit_head_split_fwd": (c_i, [c_p, c_l, c_i, c_p, c_i, c_i, c_l, c_i, c_p]),
    "mvit_head_split_bwd": (c_i, [c_p, c_p, c_l, c_i, c_i, c_i, c_l, c_i, c_p]),
    "mvit_window_preprocess": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_p]),
    "mvit_layernorm_bwd_workspace_bytes": (c_l, [c_i]),
    "mvit_layernorm_bwd": (c_i, [c_p, c_p, c_p, c_i, c_l, c_f, c_p, c_p, c_p, c_p, c_i, c_p, c_l, c_i, c_f, c_p, c_p, c_l, c_p]),
    "mvit_gelu_fwd": (c_i, [c_p, c_p, c_l, c_i, c_p]),
    "mvit_gelu_bwd": (c_i, [c_p, c_p, c_p, c_l, c_i, c_p]),
    "mvit_linear_gelu_fwd": (c_i, [c_p, c_l, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_p]),
    "mvit_linear_dgelu_fwd": (c_i, [c_p, c_l, c_p, c_p, c_l, c_p, c_p, c_l, c_i, c_i, c_i, c_p]),
    "mvit_linear_gelu_fwd_dsave": (c_i, [c_p, c_l, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_p]),
    "mvit_linear_dact_fwd": (c_i, [c_p, c_l, c_p, c_p, c_l, c_p, c_p, c_l, c_i, c_i, c_i, c_p]),
    "mvit_linear_wgrad_workspace_bytes": (c_l, [c_i, c_l, c_i, c_l, c_i, c_l, c_i, c_i, c_i]),
    "mvit_linear_wgrad": (c_i, [c_p, c_i, c_l, c_p, c_i, c_l, c_p, c_l, c_p, c_p, c_l, c_i, c_i, c_i, c_p, c_l, c_p]),
    "mvit_colsum_workspace_bytes": (c_l, [c_i]),
    "mvit_colsum": (c_i, [c_p, c_i, c_l, c_i, c_p, c_l, c_p, c_i, c_p, c_p]),
    "mvit_attention_bwd_workspace_bytes": (c_l, [c_i, c_i, c_i, c_i]),
    "mvit_attention_bwd": (c_i, [c_p] * 10 + [c_i, c_i, c_i, c_i, c_f, c_i, c_i, c_p]),
    "mvit_pool_bwd_workspace_bytes": (c_l, [c_i, c_i, c_i, c_i, c_i, c_i]),
    "mvit_pool_conv_ln_bwd": (c_i, [c_p, c_l, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_p]),
    "mvit_pool_conv_ln_fwd_train": (c_i, [c_p, c_l, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_p]),
    "mvit_pool_conv_ln_bwd_saved": (c_i, [c_p, c_l, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_p]),
    "mvit_pool_conv_ln_fwd_train_kv": (c_i, [c_p, c_l, c_i] + [c_p] * 9 + [c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_p]),
    "mvit_pool_conv_ln_bwd_saved_kv": (c_i, [c_p, c_l, c_i] + [c_p] * 15 + [c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "mvit_maxpool_skip_bwd": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "mvit_maxpool_skip_fwd_idx": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "mvit_maxpool_skip_bwd_idx": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "mvit_proj_maxpool_fwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "mvit_proj_maxpool_bwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "mvit_stem_bwd_workspace_bytes": (c_l, [c_i, c_i, c_i, c_i]),
    "mvit_stem_bwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_l, c_p]),
    "mvit_head_ln_partial": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_p]),
    "mvit_head_project_train": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "mvit_head_bwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "mvit_soft_ce": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_f, c_p]),
    "mvit_mt_chunk_bytes": (c_i, []),
    "mvit_grad_norm": (c_i, [c_p, c_i, c_f, c_p, c_p, c_p]),
    "mvit_adamw_step": (c_i, [c_p, c_i, c_p, c_f, c_f, c_f, c_f, c_i, c_p]),
    "mvit_adamw_step_dev": (c_i, [c_p, c_i, c_p, c_p, c_f, c_f, c_f, c_p]),
    "mvit_reduce_queue_begin": (c_i, []),
    "mvit_reduce_queue_flush": (c_i, [c_p]),
}
EXPORTS = tuple(_SIGS)


def build(force=False):
    """Compile every HIP source for gfx950 into lib/libmvit_hip.so (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "mvit_hip.h"))
    if not force and all(os.path.exists(p) and all(os.path.getmtime(p) >= os.path.getmtime(s) for s in srcs)
                         for p in (LIB_PATH, LIB_PATH_F16)):
        return LIB_PATH
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    subprocess.check_call(["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1))])
    return LIB_PATH


def _load(path):
    if not os.path.exists(path):
        raise RuntimeError(
            "%s is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C aicity_action_amd/csrc`. There is no CPU fallback for the HIP path." % path)
    # torch first: it ships its own libamdhip64; if this library were loaded before torch, the process would end up with two HIP
    # runtimes (the system one resolved for us, the bundled one for torch) and every launch on torch-allocated memory would fail
    import torch  # noqa: F401
    L = ctypes.CDLL(path)
    for name, (res, args) in _SIGS.items():
        fn = getattr(L, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    return L


def lib(half="bf16"):
    """The kernel library; half selects what the 16-bit activation type (dtype code BF16) means: "bf16" or "fp16"."""
    global _lib, _lib_f16
    if half == "fp16":
        if _lib_f16 is None:
            _lib_f16 = _load(LIB_PATH_F16)
        return _lib_f16
    if _lib is None:
        _lib = _load(LIB_PATH)
    return _lib


# bench.py's in-step roofline: when this is a list, every fused-attention launch of the model is bracketed by two timing events on
# its launch stream and (kind, algorithmic FLOPs, event, event) is appended -- what the kernels cost INSIDE a step, between the
# step's other launches and beside its side streams.  None (the default) = no events, no overhead.
ATT_TIMER = None


def _att_timed(kind, flops, fn):
    if ATT_TIMER is None:
        return fn()
    import torch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn()
    e1.record()
    ATT_TIMER.append((kind, flops, e0, e1))
    return rc


def attention_fwd(L, q, k, v, o, lse, B, h, Lq, Lk, scale, add_q, act, st):
    """mvit_attention_fwd on torch tensors.  (A key-split form of the ragged last query tile was measured in round 4 -- stage-3 forward
    168.2 -> 158.9 us at B = 8 but 77.5 -> 96 us for the 3-clip sub-batches of the inference path, and the choice may not depend on the
    batch -- and left the library in round 5: tools/probes/attn_fwd_keysplit.patch, profiles/r4_attn_tail.txt.)"""
    return _att_timed("fwd", 4.0 * B * h * Lq * Lk * 96,
                      lambda: L.mvit_attention_fwd(ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse), B, h, Lq, Lk, scale, add_q, act, st))


def attention_bwd(L, q, k, v, o, lse, d_o, dq, dk, dv, ws, B, h, Lq, Lk, scale, add_q, act, st):
    """mvit_attention_bwd on torch tensors (delta / dQ pass / dK,dV pass: everything is ordered on `st` again when it returns).  Credited
    FLOPs = 2 x the forward's (the recomputed products are not credited)."""
    return _att_timed("bwd", 8.0 * B * h * Lq * Lk * 96,
                      lambda: L.mvit_attention_bwd(ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse), ptr(d_o), ptr(dq), ptr(dk), ptr(dv), ptr(ws),
                                                   B, h, Lq, Lk, scale, add_q, act, st))


def check(rc, what=""):
    if rc != 0:
        msg = lib().mvit_strerror(rc).decode()
        raise RuntimeError("mvit HIP kernel %s failed: %s (code %d)" % (what, msg, rc))


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_STREAMS = {}


def shared_streams(dev, n, tag="sub"):
    """Process-wide torch side streams of a device (n of them under `tag`).  Every model used to create its own pair of
    sub-batch streams; HIP multiplexes streams onto a handful of hardware queues, so the second model's two streams could land on
    ONE queue and its sub-batches ran one after the other (17.5 instead of 14.5 ms per forward, tools/fwd_second_model_probe.py).
    One set per process and device keeps the stream -> queue assignment of the first, working, set."""
    import torch
    key = (torch.device(dev).index, tag)
    have = _STREAMS.setdefault(key, [])
    while len(have) < n:
        have.append(torch.cuda.Stream(device=dev))
    return have[:n]
