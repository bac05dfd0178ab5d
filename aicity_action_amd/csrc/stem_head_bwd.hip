// Backward of the two ends of the network.
//   stem : dW[96][3][3][7][7], dpos_spatial, dpos_temporal from d_x0 (bias grad = column sum, see mvit_colsum)
//   head : training forward variant that also returns the (dropout-masked) pooled feature z, and the tiny
//          backward through Linear(C, num_classes) + dropout + token mean (the final LayerNorm backward is
//          mvit_layernorm_bwd in broadcast mode).
#include "common.h"

// ------------------------------------------------------------------------------------------------
// stem weight gradient (v1, fp32 VALU): persistent workgroups walk 8x8-token tiles; thread = (channel, group of
// 10 taps) and keeps its 9 planes x 10 taps partial sums in registers; they go to the workgroup's own slab at the end.
// ------------------------------------------------------------------------------------------------
#define SB_PW 35
__global__ __launch_bounds__(512) void stem_wgrad_kernel(const float* __restrict__ clip, const float* __restrict__ dx,
                                                         float* __restrict__ dW, int B, int T, int S, int To, int So,
                                                         int tiles_x, int tiles_y, float* __restrict__ part) {
    __shared__ float patch[SB_PW * SB_PW];
    __shared__ __attribute__((aligned(16))) float dt_tile[64 * 96];
    const int tid = threadIdx.x;
    const int c = tid % 96, tg = tid / 96;
    const bool on = tg < 5;
    const int tap0 = tg * 10;
    const int ntap = !on ? 0 : (tap0 + 10 <= 49 ? 10 : 49 - tap0);
    int poff[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const int tap = tap0 + k < 49 ? tap0 + k : 48;
        poff[k] = (tap / 7) * SB_PW + (tap % 7);
    }
    float acc[9][10];
#pragma unroll
    for (int p = 0; p < 9; ++p)
#pragma unroll
        for (int k = 0; k < 10; ++k) acc[p][k] = 0.f;
    const int tiles_per_frame = tiles_x * tiles_y;
    const int ntiles = B * To * tiles_per_frame;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / (To * tiles_per_frame);
        int rem = tile - b * (To * tiles_per_frame);
        const int to = rem / tiles_per_frame;
        rem -= to * tiles_per_frame;
        const int ty0 = (rem / tiles_x) * 8, tx0 = (rem % tiles_x) * 8;
        __syncthreads();
        for (int i = tid; i < 64 * 24; i += 512) {
            const int tok = i / 24, c4 = i - tok * 24;
            const int yo = ty0 + (tok >> 3), xo = tx0 + (tok & 7);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (yo < So && xo < So) v = load4(dx + (((int64_t)b * To + to) * So * So + (int64_t)yo * So + xo) * 96 + 4 * c4);
            *reinterpret_cast<float4*>(dt_tile + tok * 96 + 4 * c4) = v;
        }
#pragma unroll
        for (int p = 0; p < 9; ++p) {
            const int ci = p / 3, dt = p % 3;
            const int ti = 2 * to + dt - 1;
            __syncthreads();
            const bool t_ok = ti >= 0 && ti < T;
            for (int i = tid; i < SB_PW * SB_PW; i += 512) {
                const int py = i / SB_PW, px = i - py * SB_PW;
                const int yi = 4 * ty0 + py - 3, xi = 4 * tx0 + px - 3;
                float v = 0.f;
                if (t_ok && yi >= 0 && yi < S && xi >= 0 && xi < S) v = clip[((((int64_t)b * 3 + ci) * T + ti) * S + yi) * S + xi];
                patch[i] = v;
            }
            __syncthreads();
            if (on && t_ok) {
                for (int tok = 0; tok < 64; ++tok) {
                    const float d = dt_tile[tok * 96 + c];
                    const int base = (4 * (tok >> 3)) * SB_PW + 4 * (tok & 7);
#pragma unroll
                    for (int k = 0; k < 10; ++k) acc[p][k] = fmaf(d, patch[base + poff[k]], acc[p][k]);
                }
            }
        }
    }
    if (on) {      // part: this workgroup's own [96][441] slab (plain stores; slab_sum_kernel adds them in workgroup order)
        float* o = part + (int64_t)blockIdx.x * (96 * 441);
#pragma unroll
        for (int p = 0; p < 9; ++p)
#pragma unroll
            for (int k = 0; k < 10; ++k)
                if (k < ntap) {
                    o[(int64_t)c * 441 + p * 49 + tap0 + k] = acc[p][k];
                }
    }
}

// out[i] += sum_p part[p][i], p in index order (float4 per thread): second stage of the slab form of the stem weight gradient
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ part, int nparts, int64_t stride, float* __restrict__ out,
                                                       int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 s = *reinterpret_cast<const float4*>(part + 4 * i);
#pragma unroll 8
        for (int p = 1; p < nparts; ++p) {                 // (fixed order; unrolled so that 8 loads are in flight)
            const float4 t = *reinterpret_cast<const float4*>(part + p * stride + 4 * i);
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        float4 cur = *reinterpret_cast<const float4*>(out + 4 * i);
        cur.x += s.x; cur.y += s.y; cur.z += s.z; cur.w += s.w;
        *reinterpret_cast<float4*>(out + 4 * i) = cur;
    }
}

// ------------------------------------------------------------------------------------------------
// stem weight gradient on the matrix cores (16-bit path):  dW[c][ci,dt,dy,dx] = sum_tok dxg[tok][c] * in[ci][2t-1+dt][4y-3+dy][4x-3+dx]
// as a GEMM whose contraction runs over the tokens of one output row (b, t, y).  Workgroup = 3 waves (wave = dt) working
// for one input channel ci; per output row it stages
//   * the row's token gradients [So][96] as a 16-bit row-major tile (A operand by transposing reads), and
//   * the 3 x 7 input rows it needs as "dx planes"  P[dt][dy][dx][x] = in[4x-3+dx]  (each fp32 pixel goes to one or two
//     planes), so that the B fragment of 8 consecutive tokens for one (dy,dx) column is one aligned 16-byte read;
//     plane rows are 240 B apart -> the 16 lanes of a ds_read_b128 hit 16 distinct 16-B slots.
// Columns: k-block kb (0,1), lane column j -> dy = 4kb + (j>>3), dx = j&7; dy = 7 / dx = 7 are zero padding.
// Accumulators (3 c-blocks x 2 k-blocks per wave) live across all rows of the persistent workgroup; one fp32 atomic per
// element at the end.
// ------------------------------------------------------------------------------------------------
#define SW_XS 120                       // plane row stride in elements (240 B)
#define SW_PLANE (8 * SW_XS)            // one (dt,dy): 8 dx rows
#define SW_PBYTES (3 * 7 * SW_PLANE * 2)    // 40320
#define SW_DROWB 192
#define SW_DBYTES (112 * SW_DROWB)          // 21504
typedef __attribute__((address_space(3))) bf16x4 sw_lds_b4;

__global__ __launch_bounds__(192) void stem_wgrad_mfma_kernel(const float* __restrict__ clip, const float* __restrict__ dxg,
                                                              float* __restrict__ dW, int B, int T, int S, int To, int So,
                                                              float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) char smem[SW_PBYTES + SW_DBYTES];
    bf16_t* sP = reinterpret_cast<bf16_t*>(smem);
    char* sD = smem + SW_PBYTES;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    // Workgroup -> (row group, input channel): the three channels of a group sit on ONE XCD (ids xcd + 8 (3k + ci): blockIdx.x % 8 picks
    // the XCD), so the token gradients they all read come from HBM once and from that XCD's L2 twice; a group walks a CONTIGUOUS run of
    // output rows, so the 3 of 7 input rows that consecutive output rows share are L2 hits as well.
    int grp, ci;
    {
        const int nwg0 = gridDim.x / 3;
        if ((nwg0 & 7) == 0) {
            const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
            grp = (slot / 3) * 8 + xcd;
            ci = slot % 3;
        } else {
            grp = blockIdx.x / 3;
            ci = blockIdx.x % 3;
        }
    }
    const int nks = (So + 15) / 16;
    for (int i = tid; i < (SW_PBYTES + SW_DBYTES) / 16; i += 192) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);

    f32x16 acc[3][2];
#pragma unroll
    for (int cb = 0; cb < 3; ++cb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[cb][kb][i] = 0.f;

    // fragment addressing
    const int i16 = lane & 15, gi = lane >> 4;
    const int a_lane = (8 * h + (i16 >> 2)) * SW_DROWB + (16 * (gi & 1) + 4 * (i16 & 3)) * 2;     // + ks*16 rows, + cb*64 B
    int b_lane[2];
    bool b_ok[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int dy = 4 * kb + (r >> 3), dx = r & 7;
        b_ok[kb] = dy < 7 && dx < 7;
        b_lane[kb] = ((wave * 7 + (dy < 7 ? dy : 0)) * SW_PLANE + dx * SW_XS + 8 * h) * 2;    // + ks*16 elements
    }

    const int nrows = B * To * So;        // output rows (b, to, yo)
    const int nwg = gridDim.x / 3;
    const int x4 = S / 4;                 // float4 groups per input row == So
    const int per = nrows / nwg, rem = nrows - per * nwg;
    const int row0 = grp * per + (grp < rem ? grp : rem), row1 = row0 + per + (grp < rem ? 1 : 0);
    for (int row = row0; row < row1; ++row) {
        const int yo = row % So;
        const int to = (row / So) % To;
        const int b = row / (So * To);
        __syncthreads();                  // everyone is done with the previous row's tiles
        // ---- token gradients of the row -> 16-bit [So][96] ----------------------------------------
        const float* drow = dxg + (int64_t)row * So * 96;
        for (int i = tid; i < So * 12; i += 192) {
            const int tok = i / 12, c8 = i - tok * 12;
            float4 lo, hi;
            load8(drow + tok * 96 + 8 * c8, lo, hi);
            uint4 o;
            o.x = pack_bf16x2(lo.x, lo.y); o.y = pack_bf16x2(lo.z, lo.w); o.z = pack_bf16x2(hi.x, hi.y); o.w = pack_bf16x2(hi.z, hi.w);
            *reinterpret_cast<uint4*>(sD + tok * SW_DROWB + c8 * 16) = o;
        }
        // ---- input rows -> dx planes -----------------------------------------------------------------
        for (int i = tid; i < 21 * x4; i += 192) {
            const int pr = i / x4, xq = i - pr * x4;        // plane row (dt*7+dy), float4 group x' of the input row
            const int dt = pr / 7, dy = pr - dt * 7;
            const int ti = 2 * to + dt - 1, yi = 4 * yo + dy - 3;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ti >= 0 && ti < T && yi >= 0 && yi < S)
                v = load4(clip + ((((int64_t)b * 3 + ci) * T + ti) * S + yi) * S + 4 * xq);
            bf16_t* pl = sP + pr * SW_PLANE;
            // in[4x'+p]: plane dx = p+3 at x = x'; plane dx = p-1 at x = x'+1 (p >= 1)
            pl[3 * SW_XS + xq] = f32_to_bf16(v.x);
            pl[4 * SW_XS + xq] = f32_to_bf16(v.y);
            pl[5 * SW_XS + xq] = f32_to_bf16(v.z);
            pl[6 * SW_XS + xq] = f32_to_bf16(v.w);
            if (xq + 1 < So) {
                pl[0 * SW_XS + xq + 1] = f32_to_bf16(v.y);
                pl[1 * SW_XS + xq + 1] = f32_to_bf16(v.z);
                pl[2 * SW_XS + xq + 1] = f32_to_bf16(v.w);
            }
        }
        __syncthreads();
        const int ti_w = 2 * to + wave - 1;
        if (ti_w < 0 || ti_w >= T) continue;          // this wave's temporal tap reads padding only (uniform per wave)
        for (int ks = 0; ks < nks; ++ks) {
            bf16x8 af[3], bf[2];
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) {
                const char* ap = sD + a_lane + ks * 16 * SW_DROWB + cb * 64;
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sw_lds_b4*)(ap));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sw_lds_b4*)(ap + 4 * SW_DROWB));
                af[cb][0] = lo[0]; af[cb][1] = lo[1]; af[cb][2] = lo[2]; af[cb][3] = lo[3];
                af[cb][4] = hi[0]; af[cb][5] = hi[1]; af[cb][6] = hi[2]; af[cb][7] = hi[3];
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                bf[kb] = *reinterpret_cast<const bf16x8*>(smem + b_lane[kb] + ks * 32);
                if (!b_ok[kb]) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) bf[kb][e] = 0;
                }
            }
#pragma unroll
            for (int cb = 0; cb < 3; ++cb)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) acc[cb][kb] = mfma16(af[cb], bf[kb], acc[cb][kb]);
        }
    }
    // D[c][k]: row c = 32cb + (i&3) + 8(i>>2) + 4h, column k = lane&31 -> (dy, dx)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int dy = 4 * kb + (r >> 3), dx = r & 7;
        if (dy < 7 && dx < 7) {
#pragma unroll
            for (int cb = 0; cb < 3; ++cb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int c = 32 * cb + (i & 3) + 8 * (i >> 2) + 4 * h;
                    // part: slab of this row group (the three input-channel workgroups of a group write disjoint columns)
                    part[(int64_t)grp * (96 * 441) + (int64_t)c * 441 + ci * 147 + wave * 49 + dy * 7 + dx] = acc[cb][kb][i];
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// v2 (round 4) of the matrix-core weight gradient: the same GEMM (contraction over the tokens of an output row), re-staged.
//   The r2 kernel above spends its time staging, not on memory (profiles/r4_stem_ring.txt: moving its workgroups so that the token
//   gradients come from HBM once changed nothing): every pixel is written to one or two of 8 "dx planes" with 2-byte LDS stores, the
//   token gradients are converted three times (once per input-channel workgroup), and all of it waits for its loads with the MFMAs idle.
//   Here one workgroup (6 waves) owns ALL 18 column blocks of dW -- wave w: blocks 3w .. 3w+2 of (plane = ci*3+dt, dy half), 9 accumulator
//   tiles -- and walks a contiguous run of output rows (b, to, yo):
//   * the B operand comes from LINEAR 16-bit pixel rows [plane][row slot][4 zero pixels | S pixels]: a token's 8 columns are pixels
//     4x-4 .. 4x+3 (column 0 has no weight: dx = column - 1), i.e. 16 contiguous bytes at byte 8x, and ds_read_tr16_b64 -- the read that
//     already transposes the token gradients into A fragments -- takes one address per lane, so tokens 8 bytes apart are as good as
//     a dense matrix.  Each pixel is converted and written once (8-byte stores);
//   * consecutive output rows share 3 of their 7 input rows: a ring of 8 row slots per plane (slot = yi & 7), 4 new rows per step;
//   * the token gradients of the NEXT row travel by LDS-DMA (a 43 KB linear copy) into an fp32 staging area and the next 4 x 9 pixel rows
//     into registers while the MFMAs of the current row run; two barriers per row.
// ------------------------------------------------------------------------------------------------
#define SG_NT 384
#define SG_PXB 912                         // bytes of a pixel row: 4 + 448 + 4 pixels of 16 bit
#define SG_RING (9 * 8 * SG_PXB)           // 65,664
#define SG_DROWB 192
#define SG_SD (112 * SG_DROWB)             // 21,504: token gradients of the row, 16 bit [token][96]
#define SG_STG (112 * 384)                 // 43,008: the next row's token gradients, fp32, as they lie in memory
#define SG_LDS (SG_RING + SG_SD + SG_STG)  // 130,176

__global__ __launch_bounds__(SG_NT, 1) void stem_wgrad_rows_kernel(const float* __restrict__ clip, const float* __restrict__ dxg, int B, int T, int S,
                                                                   int To, int So, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ring = smem;
    char* const sD = smem + SG_RING;
    char* const stg = smem + SG_RING + SG_SD;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int nks = So / 16 + ((So & 15) ? 1 : 0);
    const int x4 = S / 4;                                          // quads per pixel row == So
    for (int i = tid; i < SG_LDS / 16; i += SG_NT) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);

    const int nrows = B * To * So;                                 // output rows (b, to, yo)
    const int G = gridDim.x;
    const int per = nrows / G, rem = nrows - per * G;
    const int grp = blockIdx.x;
    const int row0 = grp * per + (grp < rem ? grp : rem), row1 = row0 + per + (grp < rem ? 1 : 0);

    f32x16 acc[3][3];                                              // [c block][column block of this wave]
#pragma unroll
    for (int cb = 0; cb < 3; ++cb)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[cb][j][e] = 0.f;

    // fragment addressing (ds_read_tr16_b64: the lane names 4 consecutive columns of one row; a 16-lane group = 4 rows x 16 columns)
    const int i16 = lane & 15, gi = lane >> 4;
    const int a_lane = (8 * h + (i16 >> 2)) * SG_DROWB + (16 * (gi & 1) + 4 * (i16 & 3)) * 2;       // + ks * 16 rows, + cb * 64 B; hi: + 4 rows
    const int cidx = 4 * (gi & 1) + (i16 & 3);                     // 4-column chunk of the 32 columns: dy = 4 kb + (cidx >> 1), pixels 4 (cidx & 1) ..
    const int b_lane = (8 * h + (i16 >> 2)) * 8 + (cidx & 1) * 8;  // + ks * 128; hi: + 32
    const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)(smem);

    auto row_pos = [&](int row, int& b, int& to, int& yo) {
        yo = row % So;
        to = (row / So) % To;
        b = row / (So * To);
    };
    // token gradients of `row` -> fp32 staging: a linear copy of So * 384 bytes, 1 KiB per wave instruction
    auto dma_tokens = [&](int row) {
        const char* src = reinterpret_cast<const char*>(dxg + (int64_t)row * So * 96);
        const int nchunk = So * 384 / 1024;
        const uint32_t l16 = 16u * lane;
        for (int c = wave; c < nchunk; c += 6) {
            const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds0 + SG_RING + SG_SD + (uint32_t)c * 1024u);
            const char* sc = src + (int64_t)c * 1024;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(l16), "s"(sc) : "memory");
        }
    };
    // the 4 new pixel rows (yi = 4 yo .. 4 yo + 3) of the 9 planes: wave w takes plane-rows w, w + 6, .. (plane = pr / 4, row = pr % 4),
    // a lane the quads lane and lane + 64 of a row
    auto px_load = [&](float4 (&pv)[6][2], int b, int to, int yo) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int pr = wave + 6 * k, pl = pr >> 2, nr = pr & 3;
            const int ci = pl / 3, dt = pl - 3 * ci;
            const int ti = 2 * to + dt - 1, yi = 4 * yo + nr;
            const bool ok = ti >= 0 && ti < T && yi < S;
            const float* src = clip + ((((int64_t)b * 3 + ci) * T + (ok ? ti : 0)) * S + (ok ? yi : 0)) * S;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int xq = lane + 64 * q;
                pv[k][q] = load4(src + 4 * (xq < x4 ? xq : 0));
            }
        }
    };
    auto px_write = [&](const float4 (&pv)[6][2], int to, int yo) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int pr = wave + 6 * k, pl = pr >> 2, nr = pr & 3;
            const int dt = pl % 3;
            const int ti = 2 * to + dt - 1, yi = 4 * yo + nr;
            const bool ok = ti >= 0 && ti < T && yi < S;
            char* dst = ring + (pl * 8 + (yi & 7)) * SG_PXB + 8;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int xq = lane + 64 * q;
                uint2 o = make_uint2(pack_bf16x2(pv[k][q].x, pv[k][q].y), pack_bf16x2(pv[k][q].z, pv[k][q].w));
                if (!ok) o = make_uint2(0u, 0u);
                if (xq < x4) *reinterpret_cast<uint2*>(dst + 8 * xq) = o;
            }
        }
    };
    // rows yi = 4 yo - 3 .. 4 yo - 1 (dy = 0 .. 2) of the 9 planes, synchronously: at the head of a run; a row with yo = 0 has them
    // outside the image (zero rows)
    auto px_head = [&](int b, int to, int yo) {
        for (int pr = wave; pr < 27; pr += 6) {
            const int pl = pr / 3, nr = pr - 3 * pl;
            const int ci = pl / 3, dt = pl - 3 * ci;
            const int ti = 2 * to + dt - 1, yi = 4 * yo - 3 + nr;
            const bool ok = ti >= 0 && ti < T && yi >= 0 && yi < S;
            const float* src = clip + ((((int64_t)b * 3 + ci) * T + (ok ? ti : 0)) * S + (ok ? yi : 0)) * S;
            char* dst = ring + (pl * 8 + ((yi + 8) & 7)) * SG_PXB + 8;
            for (int xq = lane; xq < x4; xq += 64) {
                const float4 v = load4(src + 4 * xq);
                uint2 o = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
                if (!ok) o = make_uint2(0u, 0u);
                *reinterpret_cast<uint2*>(dst + 8 * xq) = o;
            }
        }
    };

    if (row0 < row1) {
        int b, to, yo;
        row_pos(row0, b, to, yo);
        float4 pv[6][2];
        __syncthreads();                                           // LDS is zero
        dma_tokens(row0);
        px_load(pv, b, to, yo);
        if (yo > 0) px_head(b, to, yo);
        for (int row = row0; row < row1; ++row) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's share of the copy has landed
            __syncthreads();                                       // #1: all of it has; the previous row's MFMAs are done with sD and the ring
            // ---- staging -> 16-bit [token][96]; the new pixel rows -> ring ---------------------------------------------------------------
            for (int i = tid; i < So * 12; i += SG_NT) {
                const int tok = i / 12, c8 = i - tok * 12;
                const float4 lo = *reinterpret_cast<const float4*>(stg + tok * 384 + c8 * 32);
                const float4 hi = *reinterpret_cast<const float4*>(stg + tok * 384 + c8 * 32 + 16);
                uint4 o;
                o.x = pack_bf16x2(lo.x, lo.y); o.y = pack_bf16x2(lo.z, lo.w); o.z = pack_bf16x2(hi.x, hi.y); o.w = pack_bf16x2(hi.z, hi.w);
                *reinterpret_cast<uint4*>(sD + tok * SG_DROWB + c8 * 16) = o;
            }
            if (yo == 0 && row != row0) {                          // a new frame: rows -3 .. -1 (slots 5, 6, 7) lie above the image
                for (int i = tid; i < 27 * (SG_PXB / 16); i += SG_NT) {
                    const int pr = i / (SG_PXB / 16), pc = i - pr * (SG_PXB / 16);
                    *reinterpret_cast<uint4*>(ring + ((pr / 3) * 8 + 5 + pr % 3) * SG_PXB + 16 * pc) = make_uint4(0, 0, 0, 0);
                }
            }
            px_write(pv, to, yo);
            __syncthreads();                                       // #2: operands complete, staging area free
            const int cyo = yo;                                    // (to / yo move on to the next row below)
            if (row + 1 < row1) {                                  // the next row's data: under the MFMAs below
                dma_tokens(row + 1);
                row_pos(row + 1, b, to, yo);
                px_load(pv, b, to, yo);
            }
            // ---- dW += dxg^T . patches over the row's tokens -------------------------------------------------------------------------------
            int baddr[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int nb = 3 * wave + j, pl = nb >> 1, kb = nb & 1;
                int dy = 4 * kb + (cidx >> 1);
                dy = dy < 7 ? dy : 6;                              // (column block 1, dy = 7: padding columns, dropped at the end)
                baddr[j] = (pl * 8 + ((4 * cyo - 3 + dy + 8) & 7)) * SG_PXB + b_lane;
            }
            for (int ks = 0; ks < nks; ++ks) {
                bf16x8 af[3], bf[3];
#pragma unroll
                for (int cb = 0; cb < 3; ++cb) {
                    const char* ap = sD + a_lane + ks * 16 * SG_DROWB + cb * 64;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sw_lds_b4*)(ap));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sw_lds_b4*)(ap + 4 * SG_DROWB));
                    af[cb][0] = lo[0]; af[cb][1] = lo[1]; af[cb][2] = lo[2]; af[cb][3] = lo[3];
                    af[cb][4] = hi[0]; af[cb][5] = hi[1]; af[cb][6] = hi[2]; af[cb][7] = hi[3];
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const char* bp = ring + baddr[j] + ks * 128;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sw_lds_b4*)(bp));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sw_lds_b4*)(bp + 32));
                    bf[j][0] = lo[0]; bf[j][1] = lo[1]; bf[j][2] = lo[2]; bf[j][3] = lo[3];
                    bf[j][4] = hi[0]; bf[j][5] = hi[1]; bf[j][6] = hi[2]; bf[j][7] = hi[3];
                }
#pragma unroll
                for (int cb = 0; cb < 3; ++cb)
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[cb][j] = mfma16(af[cb], bf[j], acc[cb][j]);
            }
        }
    }
    // D[c][column]: row c = 32 cb + (e & 3) + 8 (e >> 2) + 4 h; column = lane & 31 -> (dy = 4 kb + (col >> 3), dx = (col & 7) - 1)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int nb = 3 * wave + j, pl = nb >> 1, kb = nb & 1;
        const int dy = 4 * kb + (r >> 3), dx = (r & 7) - 1;
        if (dy < 7 && dx >= 0) {
#pragma unroll
            for (int cb = 0; cb < 3; ++cb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int c = 32 * cb + (e & 3) + 8 * (e >> 2) + 4 * h;
                    part[(int64_t)grp * (96 * 441) + (int64_t)c * 441 + pl * 49 + dy * 7 + dx] = acc[cb][j][e];
                }
        }
    }
}

// dpos_s[hw][c] += sum_{t,b} dx[b][t][hw][c] ;  dpos_t[t][c] += sum_{b,hw} dx[b][t][hw][c].
// One workgroup per 8 spatial positions walks all (t, b) in a fixed order: dpos_s is complete in registers (no atomics);
// the workgroup's partial of dpos_t goes to part[block][To*96] (part != NULL: summed over blocks by the library's ordered
// column reduce) or, legacy, to fp32 atomics.
__global__ __launch_bounds__(256) void stem_pos_bwd_kernel(const float* __restrict__ dx, float* __restrict__ dps,
                                                           float* __restrict__ dpt, int B, int To, int HW, float* __restrict__ part) {
    __shared__ float red[8][96];
    const int hw0 = blockIdx.x * 8;
    const int c4 = threadIdx.x % 24, r = threadIdx.x / 24;   // 10 row slots, 8 used
    const int hw = hw0 + r;
    const bool ok = r < 8 && hw < HW;
    float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < To; ++t) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok)
            for (int b = 0; b < B; ++b) {
                const float4 v = load4(dx + (((int64_t)b * To + t) * HW + hw) * 96 + 4 * c4);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        tot.x += acc.x; tot.y += acc.y; tot.z += acc.z; tot.w += acc.w;
        __syncthreads();
        if (r < 8) *reinterpret_cast<float4*>(&red[r][4 * c4]) = acc;
        __syncthreads();
        if (threadIdx.x < 96) {
            float s = 0.f;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) s += red[rr][threadIdx.x];
            part[((int64_t)blockIdx.x * To + t) * 96 + threadIdx.x] = s;
        }
    }
    if (ok) {
        float4 cur = load4(dps + (int64_t)hw * 96 + 4 * c4);
        cur.x += tot.x; cur.y += tot.y; cur.z += tot.z; cur.w += tot.w;
        *reinterpret_cast<float4*>(dps + (int64_t)hw * 96 + 4 * c4) = cur;
    }
}

// dW, dpos_s, dpos_t are ACCUMULATED into (caller zeroes them once per step).
// act_dtype selects the weight-gradient kernel: MVIT_F32 -> exact fp32 VALU kernel, MVIT_BF16 -> matrix-core kernel on
// 16-bit copies of the clip rows and token gradients (fp32 accumulation).  Positional-embedding gradients are fp32 sums.
// workspace (mvit_stem_bwd_workspace_bytes): per-workgroup partial slabs added in a fixed order -> bit-reproducible gradients;
// NULL: the workgroups meet in fp32 atomics.
static bool stem_wgrad_v2() {                        // A/B switch: MVIT_STEM_WGRAD_V1=1 runs the r2 kernel
    static const bool v1 = getenv("MVIT_STEM_WGRAD_V1") && getenv("MVIT_STEM_WGRAD_V1")[0] == '1';
    return !v1;
}
// *mfma: 0 = fp32 VALU kernel, 1 = r2 matrix-core kernel (3 workgroups per row group), 2 = the row-ring kernel (one workgroup per row group)
static int stem_wgrad_groups(int B, int T, int S, int act_dtype, int* mfma) {
    const int To = T / 2, So = S / 4;
    *mfma = act_dtype == MVIT_BF16 && So <= 112 ? 1 : 0;
    if (*mfma) {
        const int64_t nrows = (int64_t)B * To * So;
        if (stem_wgrad_v2() && (So & 7) == 0) {
            *mfma = 2;
            return (int)(nrows < 256 ? nrows : 256);
        }
        return (int)(nrows < 168 ? nrows : 168);          // x3 input channels = 504 workgroups (2 per CU by LDS); 168 = 8 x 21: whole XCDs
    }
    const int ntiles = B * To * ((So + 7) / 8) * ((So + 7) / 8);
    return ntiles < 512 ? ntiles : 512;
}
extern "C" int64_t mvit_stem_bwd_workspace_bytes(int B, int T, int S, int act_dtype) {
    if (B <= 0 || T <= 0 || S <= 0 || (T & 1) || (S & 3)) return 0;
    int mfma;
    const int groups = stem_wgrad_groups(B, T, S, act_dtype, &mfma);
    const int To = T / 2, So = S / 4;
    const int64_t pos_blocks = (So * So + 7) / 8;
    return ((int64_t)groups * 96 * 441 + pos_blocks * To * 96) * (int64_t)sizeof(float);
}
extern "C" int mvit_stem_bwd(const float* clip, const float* dx, float* dW, float* dpos_spatial, float* dpos_temporal,
                              int B, int T, int S, int act_dtype, float* workspace, int64_t workspace_bytes, void* stream) {
    if (act_dtype != MVIT_F32 && act_dtype != MVIT_BF16) return MVIT_EDTYPE;
    if (!clip || !dx || !dW || !dpos_spatial || !dpos_temporal || B <= 0 || T <= 0 || S <= 0) return MVIT_EINVAL;
    if ((T & 1) || (S & 3)) return MVIT_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
    const int To = T / 2, So = S / 4;
    int mfma;
    const int groups = stem_wgrad_groups(B, T, S, act_dtype, &mfma);
    const int pos_blocks = (So * So + 7) / 8;
    if (!workspace || workspace_bytes < mvit_stem_bwd_workspace_bytes(B, T, S, act_dtype)) return MVIT_EINVAL;
    float* wpart = workspace;                                   // per-workgroup slabs of dW, summed in workgroup order
    float* ppart = workspace + (int64_t)groups * 96 * 441;      // partial rows of dpos_temporal
    if (mfma == 2) {
        static DevFlags attr;
        DevFlag done = dev_flag(attr);
        if (!done) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_wgrad_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SG_LDS) != hipSuccess)
                return MVIT_ELAUNCH;
            done = true;
        }
        hipLaunchKernelGGL(stem_wgrad_rows_kernel, dim3(groups), dim3(SG_NT), SG_LDS, st, clip, dx, B, T, S, To, So, wpart);
    } else if (mfma) {
        hipLaunchKernelGGL(stem_wgrad_mfma_kernel, dim3(3 * groups), dim3(192), 0, st, clip, dx, dW, B, T, S, To, So, wpart);
    } else {
        const int tiles_x = (So + 7) / 8, tiles_y = (So + 7) / 8;
        hipLaunchKernelGGL(stem_wgrad_kernel, dim3(groups), dim3(512), 0, st, clip, dx, dW, B, T, S, To, So, tiles_x, tiles_y, wpart);
    }
    MVIT_LAUNCH_CHECK();
    hipLaunchKernelGGL(slab_sum_kernel, dim3(42), dim3(256), 0, st, wpart, groups, (int64_t)96 * 441, dW, (int64_t)96 * 441 / 4);
    MVIT_LAUNCH_CHECK();
    hipLaunchKernelGGL(stem_pos_bwd_kernel, dim3(pos_blocks), dim3(256), 0, st, dx, dpos_spatial, dpos_temporal, B, To, So * So, ppart);
    MVIT_LAUNCH_CHECK();
    return mvit_internal_reduce_partials(ppart, pos_blocks, To * 96, dpos_temporal, dpos_temporal, To * 96, 1, st);
}

// ------------------------------------------------------------------------------------------------
// head, training variant: z = mean_n LN(x) (stage 1 = mvit_head_fwd's partial kernel, reused through the same
// workspace layout), z *= mask (dropout, mask holds 0 or 1/(1-p)), logits = z W^T + b.  Returns z (masked).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_project_train_kernel(const float* __restrict__ part, const float* __restrict__ w,
                                                                 const float* __restrict__ bias, const float* __restrict__ mask,
                                                                 float* __restrict__ z_out, float* __restrict__ logits, int N,
                                                                 int nchunks, int C, int ncls) {
    extern __shared__ float sm[];
    float* z = sm;
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int k = 0; k < nchunks; ++k) s += part[((int64_t)b * nchunks + k) * C + c];
        s = s / (float)N;
        if (mask) s *= mask[(int64_t)b * C + c];
        z[c] = s;
        z_out[(int64_t)b * C + c] = s;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int j = wave; j < ncls; j += 4) {
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += z[c] * w[(int64_t)j * C + c];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) logits[(int64_t)b * ncls + j] = s + bias[j];
    }
}

extern "C" int mvit_head_project_train(const float* partials, const float* w_head, const float* b_head, const float* mask,
                                       float* z_out, float* logits, int B, int N, int nchunks, int C, int num_classes,
                                       void* stream) {
    if (!partials || !w_head || !b_head || !z_out || !logits || B <= 0 || N <= 0 || C <= 0) return MVIT_EINVAL;
    hipLaunchKernelGGL(head_project_train_kernel, dim3(B), dim3(256), C * sizeof(float), as_stream(stream), partials, w_head,
                       b_head, mask, z_out, logits, N, nchunks, C, num_classes);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}

// dW[j][c] = sum_b dl[b][j] z[b][c];  db[j] = sum_b dl[b][j];  dz[b][c] = mask[b][c] * sum_j dl[b][j] W[j][c]
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dl, const float* __restrict__ z,
                                                       const float* __restrict__ w, const float* __restrict__ mask,
                                                       float* __restrict__ dW, float* __restrict__ db, float* __restrict__ dz,
                                                       int B, int C, int ncls, int accumulate) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < C) {
        for (int j = 0; j < ncls; ++j) {
            float s = 0.f;
            for (int b = 0; b < B; ++b) s += dl[(int64_t)b * ncls + j] * z[(int64_t)b * C + c];
            float* o = dW + (int64_t)j * C + c;
            *o = accumulate ? *o + s : s;
        }
        for (int b = 0; b < B; ++b) {
            float s = 0.f;
            for (int j = 0; j < ncls; ++j) s += dl[(int64_t)b * ncls + j] * w[(int64_t)j * C + c];
            dz[(int64_t)b * C + c] = mask ? s * mask[(int64_t)b * C + c] : s;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < ncls) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dl[(int64_t)b * ncls + threadIdx.x];
        db[threadIdx.x] = accumulate ? db[threadIdx.x] + s : s;
    }
}

extern "C" int mvit_head_bwd(const float* dlogits, const float* z, const float* w_head, const float* mask, float* dW,
                             float* db, float* dz, int B, int C, int num_classes, int accumulate, void* stream) {
    if (!dlogits || !z || !w_head || !dW || !db || !dz || B <= 0 || C <= 0 || num_classes <= 0 || num_classes > 256)
        return MVIT_EINVAL;
    hipLaunchKernelGGL(head_bwd_kernel, dim3((C + 255) / 256), dim3(256), 0, as_stream(stream), dlogits, z, w_head, mask, dW, db,
                       dz, B, C, num_classes, accumulate);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}
