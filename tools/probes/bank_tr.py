"""Bank-conflict enumeration of the ds_read_b64_tr_b16 fragment reads of csrc/wgrad_pp.hip (32-lane halves, 8-byte slots of the 256-byte bank row)."""
# bank-conflict check of ds_read_b64_tr_b16 for the wgrad-pp images
def conflicts(addr_fn):
    worst = 0
    for half in range(2):
        slots = {}
        for l in range(32 * half, 32 * half + 32):
            a = addr_fn(l)
            assert a % 8 == 0
            s = (a // 8) % 32            # 8-byte slot within the 256-B bank row
            slots.setdefault(s, set()).add(a)
        worst = max(worst, max(len(v) for v in slots.values()))
    return worst

# image A: [64 rows][128 cols x 2 B] 256-B rows, guide (b): off = 256*row + 16*(ch ^ (((row&3)<<2)|((row>>2)&3)))
def offA(row, ch): return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)))
def tr_addr_A(l, mbase, cbase, h2):          # cbase: first column (multiple of 16); read h2 (rows +0..3 / +4..7)
    lg, l15 = l >> 4, l & 15
    q, p = l15 >> 2, l15 & 3
    row = mbase + 8 * lg + 4 * h2 + q
    ch = cbase // 8 + (p >> 1)
    return offA(row, ch) + 8 * (p & 1)
w = 0
for mbase in (0, 32):
    for cbase in range(0, 128, 16):
        for h2 in (0, 1):
            w = max(w, conflicts(lambda l: tr_addr_A(l, mbase, cbase, h2)))
print("image A (256-B rows, guide swizzle b): worst ways =", w)

# image B: [64 rows][64 cols] 128-B rows: search swizzles pos = ch ^ f(row), f from small family
import itertools
best = None
for fa in range(8):
  for fb in range(8):
    for sa in range(0, 4):
      for sb in range(0, 4):
        def f(row, fa=fa, fb=fb, sa=sa, sb=sb): return (((row >> sa) & 7) * fa ^ ((row >> sb) & 7) * fb) & 7
        def offB(row, ch): return 128 * row + 16 * (ch ^ f(row))
        def tr_addr_B(l, mbase, cbase, h2):
            lg, l15 = l >> 4, l & 15
            q, p = l15 >> 2, l15 & 3
            row = mbase + 8 * lg + 4 * h2 + q
            return offB(row, cbase // 8 + (p >> 1)) + 8 * (p & 1)
        w = 0
        for mbase in (0, 32):
            for cbase in range(0, 64, 16):
                for h2 in (0, 1):
                    w = max(w, conflicts(lambda l: tr_addr_B(l, mbase, cbase, h2)))
        if best is None or w < best[0]:
            best = (w, fa, fb, sa, sb)
print("image B (128-B rows): best", best)
