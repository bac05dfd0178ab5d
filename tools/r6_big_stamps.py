#!/usr/bin/env python3
"""Phases of linear_big_kernel's workgroups on the chip's common 100 MHz clock (library built with -DBIG_ABL=256):
start -> main loop end -> end, per workgroup, with the CU it ran on.  For every CU: how long are 0 / 1 / 2 of its resident workgroups
in their MAIN LOOP, how long in their EPILOGUE -- do the two overlap?     usage: MVIT_HIP_LIB=<stamp lib> python tools/r6_big_stamps.py [dgder|r]"""
import collections
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aicity_action_amd import _hip  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "dgder"
L = _hip.lib()
st = torch.cuda.current_stream().cuda_stream
dev = "cuda:0"
if mode == "dgder":
    M, N, K = 50176, 1536, 384
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    aux = torch.randn(M, N, device=dev).bfloat16(); y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    sc = torch.rand(8, device=dev)
    fn = lambda: _hip.check(L.mvit_linear_dact_fwd(_hip.ptr(x), K, _hip.ptr(w), _hip.ptr(sc), (M + 7) // 8, _hip.ptr(aux), _hip.ptr(y), M, N, K, _hip.BF16, st))
else:
    M, N, K = 50176, 384, 1536
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev); res = torch.randn(M, N, device=dev); y = torch.empty(M, N, device=dev)
    fn = lambda: _hip.check(L.mvit_linear_fwd(_hip.ptr(x), _hip.BF16, K, _hip.ptr(w), _hip.ptr(bias), _hip.ptr(res), N, None, 0, _hip.ptr(y), _hip.F32, N, M, N, K,
                                              _hip.EPI_BIAS | _hip.EPI_RESIDUAL, _hip.BF16, st))
for _ in range(5):
    fn()
torch.cuda.synchronize()
fn()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8192 * 8))()
L.mvit_debug_big_stamps.restype = ctypes.c_int
assert L.mvit_debug_big_stamps(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.int64)
nwg = ((M + 127) // 128) * (N // 192)
a = a[:min(nwg, 8192)]
t0, t1, t2, cu = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
base = t0.min()
span = (t2.max() - base) / 100.0
print("%s M=%d N=%d K=%d: %d workgroups stamped, launch span %.1f us on the 100 MHz clock, %d distinct CUs" % (mode, M, N, K, len(a), span, len(set(cu.tolist()))))
print("per workgroup: main loop (start -> last MFMA issued) %.2f us mean (p10 %.2f, p90 %.2f); epilogue (-> stores acknowledged) %.2f us mean (p10 %.2f, p90 %.2f)" % (
    (t1 - t0).mean() / 100, np.percentile(t1 - t0, 10) / 100, np.percentile(t1 - t0, 90) / 100, (t2 - t1).mean() / 100, np.percentile(t2 - t1, 10) / 100, np.percentile(t2 - t1, 90) / 100))
ph = a[:, 4:8].astype(np.float64)
if ph.sum() > 0:
    nkt = (K + 63) // 64
    print("wave 0 of a workgroup, shader cycles per K-tile (%d K-tiles): wait for the slab's DMA %.0f, barrier %.0f, issue of the next slab's %d LDS-DMA pieces (+ operand prefetch) %.0f, "
          "fragment reads + 24 MFMAs %.0f  (sum %.0f)" % (nkt, ph[:, 0].mean() / nkt, ph[:, 1].mean() / nkt, 10, ph[:, 2].mean() / nkt, ph[:, 3].mean() / nkt, ph.sum(1).mean() / nkt))
# per CU occupancy of the two phases over the launch span (10 ns ticks)
T = int(t2.max() - base) + 1
tot = collections.Counter()
for c in set(cu.tolist()):
    m = np.zeros(T + 1, np.int16); e = np.zeros(T + 1, np.int16)
    for i in np.nonzero(cu == c)[0]:
        m[t0[i] - base] += 1; m[t1[i] - base] -= 1
        e[t1[i] - base] += 1; e[t2[i] - base] -= 1
    m = np.cumsum(m)[:T]; e = np.cumsum(e)[:T]
    for km in range(3):
        for ke in range(3):
            tot[(km, ke)] += int(((m == km) & (e == ke)).sum())
allt = float(sum(tot.values()))
print("share of CU time by (workgroups in main loop, workgroups in epilogue):")
for k in sorted(tot):
    if tot[k]:
        print("   main %d / epilogue %d : %5.1f %%" % (k[0], k[1], 100.0 * tot[k] / allt))
# phase agreement across the chip: how many workgroups are in the epilogue at each instant
e_all = np.zeros(T + 1, np.int32); m_all = np.zeros(T + 1, np.int32)
np.add.at(e_all, t1 - base, 1); np.add.at(e_all, t2 - base, -1)
np.add.at(m_all, t0 - base, 1); np.add.at(m_all, t1 - base, -1)
e_all = np.cumsum(e_all)[:T]; m_all = np.cumsum(m_all)[:T]
mid = slice(T // 10, 9 * T // 10)
print("chip-wide, middle 80 %% of the launch: workgroups in main loop %.0f +- %.0f, in epilogue %.0f +- %.0f (of %d resident slots)" % (
    m_all[mid].mean(), m_all[mid].std(), e_all[mid].mean(), e_all[mid].std(), 512))
