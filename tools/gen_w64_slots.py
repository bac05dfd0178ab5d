#!/usr/bin/env python3
"""Generates the per-slot softmax statements of csrc/attention_w64.hip (between the GENERATED markers).

A pair of scores goes through three ops -- B0 / B1: the two v_exp_f32 (8 issue cycles each), C: row sums + pack (two v_add 8 + v_cvt_pk 4).
The 16 pairs of a query block are software-pipelined (no op follows the op it depends on) and the op stream is cut into the MFMA
slots of a phase by issue cost.
usage: tools/gen_w64_slots.py  (rewrites the file in place)"""
import os, re
COST = {"B0": 8, "B1": 8, "C": 12}

def stream(pairs):
    """pairs: list of (query block, pair index, position) in execution order -> software-pipelined op list [(op, block, pair, position)].
    The scores come out of the MFMA chain already relative to the row's reference point and in log2 units (Q is pre-scaled, the
    accumulators start at -reference), so a pair is two exponentials and one sum + pack; C of pair r-1 sits three ops behind its B1."""
    ops, n = [], len(pairs)
    for r in range(n + 1):
        if r < n: ops.append(("B0",) + pairs[r])
        if r < n: ops.append(("B1",) + pairs[r])
        if 0 <= r - 1 < n: ops.append(("C",) + pairs[r - 1])
    return ops

def cut(ops, slots):
    """ops -> len(slots) lists, cumulative cost balanced"""
    total = sum(COST[o[0]] for o in ops)
    out, acc, k = [[] for _ in slots], 0.0, 0
    for o in ops:
        while k < len(slots) - 1 and acc + COST[o[0]] / 2 > total * (k + 1) / len(slots): k += 1
        out[k].append(o); acc += COST[o[0]]
    return out

def emit(ops, s, pf, ps):
    """in-flight state is indexed by position in the pair list modulo 2 (two pairs in flight)"""
    t = []
    for o, j, e, k in ops:
        E, K, J = "IC<%d>{}" % e, "IC<%d>{}" % (k % 2), "J%d{}" % j
        if o == "B0": t.append("ex_b0(%s, %s, %s, %s);" % (s, J, E, K))
        elif o == "B1": t.append("ex_b1(%s, %s, %s, %s);" % (s, J, E, K))
        else: t.append("ex_c(%s, %s, %s, %s, %s[%d]);" % (pf, J, E, K, ps, j))
    return " ".join(t)

def body(first, last, pairs, s, pf, ps, ind):
    slots = list(range(first, last + 1))
    pairs = [(j, e, k) for k, (j, e) in enumerate(pairs)]
    parts = cut(stream(pairs), slots)
    lines = []
    for sl, ops in zip(slots, parts):
        if ops: lines.append("%sif constexpr (I == %d) { %s }" % (ind, sl, emit(ops, s, pf, ps)))
    print("slots %d..%d issue cycles:" % (first, last), [sum(COST[o[0]] for o in p) for p in parts])
    return "\n".join(lines) + "\n"

# phase 1 of step t: query block 1 of tile t; phase 2: block 0 of tile t+1 (behind its row maxima).  Measured alternative: the last
# four pairs of block 0 deferred to the next phase 1 (even issue load, 30 / 29 cycles per slot) -- phase 1, which also carries the
# V^T fragment reads and the LDS-DMA pieces, grew by more than phase 2 shrank (2691 against 2662 cycles per tile).
P1 = [(1, e) for e in range(16)]
P2 = [(0, e) for e in range(16)]
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "aicity_action_amd", "csrc", "attention_w64.hip")
src = open(path).read()
for tag, txt in (("PHASE1", body(0, 23, P1, "so", "pc", "psA", " " * 16)), ("PHASE2", body(7, 23, P2, "sn", "pn", "psB", " " * 16))):
    a = src.index("// GENERATED %s BEGIN" % tag); a = src.index("\n", a) + 1
    b = src.index("                // GENERATED %s END" % tag)
    src = src[:a] + txt + src[b:]
open(path, "w").write(src)
