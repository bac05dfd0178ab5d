"""CPU sizing probe for a matrix-core form of the pooling convs (attention.py:12-83): im2col on the fly from a token-major LDS image.

D^T[16 channels][16 positions] += A[16 ch][K = 32] . B[K][16 pos] with K = 2 taps x 16 channels: A = the taps' weights on the channel
diagonal (constant per wave, in registers), B = lane (n = position, kq): 8 consecutive channels of position p_n + tap(kq) -- one
ds_read_b128 from the LDS image [position][96 channels x 2 B].  27 taps = 14 k-steps per (16 positions x 16 channels).  The instruction
count is fine (1.05 M MFMA 16x16x32 for the stage-3 q pool = ~8 us of matrix pipe, 1 KiB of LDS per MFMA = ~8 us at 256 B/clk/CU);
the question this script answers: is there an LDS layout in which those reads are bank-conflict free?  ds_read_b128 serves 16 lanes per
cycle in the groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, +32 (MI355X_MICROARCH.md): 16 lanes x 16 B must hit 16 different 16-byte
bank groups (or the same address).  Enumerates pitches / rotations / kq -> (tap, channel half) maps over every k-step and block."""
import itertools

GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS += [[l + 32 for l in g] for g in list(GROUPS)]
XW = 30                      # plane row = 28 positions + 2 halo columns
TAPS = [(dt, dy, dx) for dt in range(3) for dy in range(3) for dx in range(3)] + [None]      # 27 + one zero tap = 14 pairs


def cycles(pitch16, rot, kqmap, pairing, row_stride=XW, plane=6 * XW):
    """LDS cycles per wave-instruction averaged over k-steps, blocks (x0 = 0, 12) and channel groups; 4.0 = conflict-free."""
    tot, n = 0, 0
    for j, (ta, tb) in enumerate(pairing):
        for x0 in (0, 12):
            for grp in range(6):
                cyc = 0
                for g in GROUPS:
                    slots = {}
                    for lane in g:
                        pos_n, kq = lane % 16, lane // 16
                        tsel, half = kqmap[kq]
                        tap = (ta, tb)[tsel]
                        if tap is None:
                            tap = (1, 1, 1)
                        p = tap[0] * plane + (1 + tap[1]) * row_stride + x0 + pos_n + tap[2]        # linear LDS position
                        chunk = 2 * grp + half
                        a16 = pitch16 * p + (chunk + rot(p)) % 12 if pitch16 == 12 else pitch16 * p + chunk
                        slots.setdefault(a16 % 16, set()).add(a16)
                    cyc += max(len(v) for v in slots.values())
                tot += cyc
                n += 1
    return tot / n


def pairings():
    flat = list(TAPS)
    yield "consecutive taps", [(flat[2 * i], flat[2 * i + 1]) for i in range(14)]
    # pairs inside one (dt, dy) row: (dx0, dx1), (dx2, zero): 18 k-steps
    p2 = []
    for dt in range(3):
        for dy in range(3):
            p2 += [((dt, dy, 0), (dt, dy, 1)), ((dt, dy, 2), None)]
    yield "x-adjacent pairs (18 k-steps)", p2


if __name__ == "__main__":
    rots = {"none": lambda p: 0, "(p>>2)&3": lambda p: (p >> 2) & 3, "p&3": lambda p: p & 3, "(p>>1)&3": lambda p: (p >> 1) & 3,
            "(p>>2)%12": lambda p: (p >> 2) % 12, "(p*5>>2)&7": lambda p: (p * 5 >> 2) & 7}
    kqmaps = {"kq=(tap,half): 0:(A,0) 1:(A,1) 2:(B,0) 3:(B,1)": [(0, 0), (0, 1), (1, 0), (1, 1)],
              "0:(A,0) 1:(B,0) 2:(A,1) 3:(B,1)": [(0, 0), (1, 0), (0, 1), (1, 1)]}
    best = []
    for pname, pairing in pairings():
        for kname, kqmap in kqmaps.items():
            for pitch16 in (12, 13, 14, 15, 17):
                for rname, rot in (rots.items() if pitch16 == 12 else [("-", rots["none"])]):
                    c = cycles(pitch16, rot, kqmap, pairing)
                    best.append((c * len(pairing), c, pname, kname, pitch16 * 16, rname))
    best.sort()
    print("LDS cycles per wave-instruction (4.0 = conflict-free), x k-steps per (16 pos x 16 ch) block = cost; best 10 of %d layouts:" % len(best))
    for cost, c, pname, kname, pitch, rname in best[:10]:
        print("  cost %6.1f  cycles/instr %.2f  pitch %3d B  rotation %-12s  %-32s  %s" % (cost, c, pitch, rname, pname, kname))
