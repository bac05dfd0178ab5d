#!/usr/bin/env python3
"""Generates the per-slot softmax statements of csrc/attention_w64.hip (between the GENERATED markers).

A pair of scores goes through four ops -- A: t = s*c - m*c (v_pk_fma, 8 issue cycles), B0 / B1: the two v_exp (8 each), C: row sum +
pack (v_pk_add 8 + v_cvt_pk 4).  The 16 pairs of a query block are software-pipelined (A of pair r beside the exps of pair r-1 and the
C of pair r-2: no op follows the op it depends on) and the resulting op stream is cut into the MFMA slots of a phase by issue cost.
usage: tools/gen_w64_slots.py  (rewrites the file in place)"""
import os, re
COST = {"A": 8, "B0": 8, "B1": 8, "C": 12}

def stream(npairs=16):
    ops = []
    for r in range(npairs + 2):
        # order inside a round: every op sits three ops behind the one it depends on
        if 0 <= r - 1 < npairs: ops.append(("B0", r - 1))
        if r < npairs: ops.append(("A", r))
        if 0 <= r - 2 < npairs: ops.append(("C", r - 2))
        if 0 <= r - 1 < npairs: ops.append(("B1", r - 1))
    return ops

def cut(ops, slots):
    """ops -> len(slots) lists, cumulative cost balanced"""
    total = sum(COST[o] for o, _ in ops)
    out, acc, k = [[] for _ in slots], 0.0, 0
    for o in ops:
        while k < len(slots) - 1 and acc + COST[o[0]] / 2 > total * (k + 1) / len(slots): k += 1
        out[k].append(o); acc += COST[o[0]]
    return out

def emit(ops, s, pf, j, ps):
    t = []
    for o, e in ops:
        E = "IC<%d>{}" % e
        if o == "A": t.append("ex_a(%s, %s{}, %s, mc2);" % (s, j, E))
        elif o == "B0": t.append("ex_b0(%s);" % E)
        elif o == "B1": t.append("ex_b1(%s);" % E)
        else: t.append("ex_c(%s, %s{}, %s, %s);" % (pf, j, E, ps))
    return " ".join(t)

def body(first, last, s, pf, j, ps, ind):
    slots = list(range(first, last + 1))
    parts = cut(stream(), slots)
    lines = []
    for sl, ops in zip(slots, parts):
        if ops: lines.append("%sif constexpr (I == %d) { %s }" % (ind, sl, emit(ops, s, pf, j, ps)))
    return "\n".join(lines) + "\n"

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "aicity_action_amd", "csrc", "attention_w64.hip")
src = open(path).read()
for tag, txt in (("PHASE1", body(0, 23, "so", "pc", "J1", "ps1", " " * 16)), ("PHASE2", body(7, 23, "sn", "pn", "J0", "ps0", " " * 16))):
    a = src.index("// GENERATED %s BEGIN" % tag); a = src.index("\n", a) + 1
    b = src.index("                // GENERATED %s END" % tag)
    src = src[:a] + txt + src[b:]
open(path, "w").write(src)
for tag, (f, l) in (("phase 1", (0, 23)), ("phase 2", (7, 23))):
    parts = cut(stream(), list(range(f, l + 1)))
    print(tag, "issue cycles per slot:", [sum(COST[o] for o, _ in p) for p in parts])
