// Cube embedding: Conv3d(3->96, k(3,7,7), s(2,4,4), p(1,3,3)) + bias + separable position embedding,
// written token-major [B][T'*H'*W'][96] fp32 (the residual stream).
//
// v1 (exact fp32 VALU): an 8x8 output-token tile per 256-thread workgroup; the (ci,dt) input plane
// patch (35x35) and its 49x96 weight slab are staged in LDS; thread = (token, 24-channel group), so
// weight reads are wave-uniform broadcasts.
#include "common.h"

#define ST_TY 8
#define ST_TX 8
#define ST_PH (4 * ST_TY + 3)  // 35
#define ST_PW (4 * ST_TX + 3)  // 35

__global__ __launch_bounds__(256) void stem_f32_kernel(const float* __restrict__ clip, const float* __restrict__ w,
                                                       const float* __restrict__ bias, const float* __restrict__ pos_s,
                                                       const float* __restrict__ pos_t, float* __restrict__ x, int T,
                                                       int S, int To, int So) {
    __shared__ float patch[ST_PH * ST_PW];
    __shared__ __attribute__((aligned(16))) float wsl[49 * 96];
    const int tiles_x = (So + ST_TX - 1) / ST_TX;
    const int tx0 = (blockIdx.x % tiles_x) * ST_TX;
    const int ty0 = (blockIdx.x / tiles_x) * ST_TY;
    const int to = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x;
    const int tok = tid & 63, cg = tid >> 6;
    const int ly = tok >> 3, lx = tok & 7;
    float acc[24];
#pragma unroll
    for (int e = 0; e < 24; ++e) acc[e] = 0.f;
    for (int ci = 0; ci < 3; ++ci)
        for (int dt = 0; dt < 3; ++dt) {
            const int ti = 2 * to + dt - 1;
            __syncthreads();
            const bool t_ok = ti >= 0 && ti < T;
            for (int i = tid; i < ST_PH * ST_PW; i += 256) {
                const int py = i / ST_PW, px = i - py * ST_PW;
                const int yi = 4 * ty0 + py - 3, xi = 4 * tx0 + px - 3;
                float v = 0.f;
                if (t_ok && yi >= 0 && yi < S && xi >= 0 && xi < S)
                    v = clip[((((int64_t)b * 3 + ci) * T + ti) * S + yi) * S + xi];
                patch[i] = v;
            }
            for (int i = tid; i < 49 * 96; i += 256) {
                const int tap = i / 96, c = i - tap * 96;
                wsl[i] = w[((c * 3 + ci) * 3 + dt) * 49 + tap];
            }
            __syncthreads();
            if (!t_ok) continue;
#pragma unroll 1
            for (int dy = 0; dy < 7; ++dy)
#pragma unroll
                for (int dx = 0; dx < 7; ++dx) {
                    const float v = patch[(4 * ly + dy) * ST_PW + 4 * lx + dx];
                    const float* wt = wsl + (dy * 7 + dx) * 96 + cg * 24;
#pragma unroll
                    for (int e = 0; e < 24; e += 4) {
                        const float4 ww = *reinterpret_cast<const float4*>(wt + e);
                        acc[e] = fmaf(v, ww.x, acc[e]); acc[e + 1] = fmaf(v, ww.y, acc[e + 1]);
                        acc[e + 2] = fmaf(v, ww.z, acc[e + 2]); acc[e + 3] = fmaf(v, ww.w, acc[e + 3]);
                    }
                }
        }
    const int yo = ty0 + ly, xo = tx0 + lx;
    if (yo < So && xo < So) {
        const int hw = yo * So + xo;
        float* orow = x + (((int64_t)b * To + to) * So * So + hw) * 96 + cg * 24;
        const float* ps = pos_s + (int64_t)hw * 96 + cg * 24;
        const float* pt = pos_t + (int64_t)to * 96 + cg * 24;
#pragma unroll
        for (int e = 0; e < 24; e += 4) {
            const float4 bb = load4(bias + cg * 24 + e), p1 = load4(ps + e), p2 = load4(pt + e);
            // reference order: (conv + bias) + (pos_s + pos_t)
            float4 v;
            v.x = (acc[e] + bb.x) + (p1.x + p2.x);
            v.y = (acc[e + 1] + bb.y) + (p1.y + p2.y);
            v.z = (acc[e + 2] + bb.z) + (p1.z + p2.z);
            v.w = (acc[e + 3] + bb.w) + (p1.w + p2.w);
            store4(orow + e, v);
        }
    }
}


// ------------------------------------------------------------------------------------------------
// bf16 MFMA path: persistent implicit GEMM.
//   K ordering k = (ci*3+dt)*7+dy) * 8 + dx  (63 tap rows + 1 zero row, dx padded 7 -> 8): K = 512 = 32
//   k-steps of 16; the A fragment of a token is then 8 CONSECUTIVE input pixels of one (ci,dt,dy) row --
//   two ds_read_b64 from a bf16 halo patch in LDS, no im2col buffer.
//   One workgroup per CU walks 1x8x16-token tiles.  Waves 0-2 each own one 32-channel block with its
//   whole 32x512 weight slab resident in 128 VGPRs (B fragments) and compute 128 tokens x 32 channels;
//   all four waves share the conversion of the NEXT tile's fp32 halo patch to bf16 into the other LDS buffer: the loads are
//   requested before the MFMA work and converted after it.  Output rows are written 128 B (32 fp32 channels) per half-wave with
//   bias + separable position embedding fused.
// ------------------------------------------------------------------------------------------------
#define SM_TY 8
#define SM_TX 16
#define SM_PH 35                       // 4*8+3 patch rows
#define SM_ROWB 144                    // 72 bf16 per patch row (68 used)
#define SM_PLANE (SM_PH * SM_ROWB)     // 5040
#define SM_PATCH (9 * SM_PLANE)        // 45360

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f4a __attribute__((ext_vector_type(4)));      // 16-byte aligned
#ifndef STEM_ABL
#define STEM_ABL 0     // timing ablations (tools): 1 no halo fill, 2 no MFMA loop, 4 no epilogue stores
#endif

__global__ __launch_bounds__(256, 1) void stem_mfma_kernel(const float* __restrict__ clip, const float* __restrict__ w,
                                                           const float* __restrict__ bias, const float* __restrict__ pos_s,
                                                           const float* __restrict__ pos_t, float* __restrict__ x, int B,
                                                           int T, int S, int To, int So, int tiles_x, int tiles_y) {
    __shared__ __attribute__((aligned(16))) char patch[2][SM_PATCH];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int tiles_per_frame = tiles_x * tiles_y;
    const int ntiles = B * To * tiles_per_frame;

    // ---- B fragments (compute waves): W[n = 32*wave + r][row63 = 2s+h][dx = j] ---------------------
    bf16x8 bfrag[32];
    if (wave < 3) {
        const float* wn = w + (int64_t)(32 * wave + r) * 441;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const int row = 2 * s + h;
            uint32_t pk[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float a = (row < 63) ? wn[row * 7 + 2 * jj] : 0.f;
                const float b2 = (row < 63 && 2 * jj + 1 < 7) ? wn[row * 7 + 2 * jj + 1] : 0.f;
                pk[jj] = pack_bf16x2(a, b2);
            }
            uint4 u = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            bfrag[s] = *reinterpret_cast<bf16x8*>(&u);
        }
    }

    // Halo patch of a tile: 9 (ci,dt) planes x 35 rows x 68 px fp32 -> bf16 in LDS.  Every wave takes every 4th group of 3 rows
    // (27 16-byte loads per lane).  The loads of the NEXT tile are requested before the MFMA work of the current one and
    // converted / written to the other LDS buffer after it, so their latency hides under the matrix work.  Every load is an
    // unconditional 16-byte load at a CLAMPED address (row and column), fixed up afterwards: with bounds tests around the loads
    // the compiler waited for each one (and a single loader wave's 105 round trips per tile were 80 % of the kernel).
    constexpr int FN = 27;
    // 16-byte ALIGNED loads (a 16-byte load at a 4-byte-aligned address is split by the memory pipeline and was 4x slower): a lane
    // loads image pixels [xa, xa+3], xa = 4*tx0 - 4 + 4*seg for 18 segments per row; the LDS piece j holds pixels xa_j+1 .. xa_j+4
    // (patch pixel 0 = image pixel 4*tx0 - 3, which keeps the MFMA waves' 8-byte fragment reads aligned), i.e. elements 1..3 of the
    // lane's quad and element 0 of the next lane's.  Rows and columns outside the image load a clamped address and are zeroed.
    const bool lane_on = lane < 54;
    const int rig = lane_on ? lane / 18 : 0, seg = lane_on ? lane - rig * 18 : 0;    // idle lanes shadow lane 0 (valid addresses)
    auto tile_pos = [&](int tile, int& b, int& to, int& ty0, int& tx0) {
        b = tile / (To * tiles_per_frame);
        int rem = tile - b * (To * tiles_per_frame);
        to = rem / tiles_per_frame;
        rem -= to * tiles_per_frame;
        ty0 = (rem / tiles_x) * SM_TY;
        tx0 = (rem % tiles_x) * SM_TX;
    };
    auto fill_issue = [&](int tile, f4u (&v)[FN], uint32_t& rmask) {
        int b, to, ty0, tx0;
        tile_pos(tile, b, to, ty0, tx0);
        const int xa = 4 * tx0 - 4 + 4 * seg;
        const bool xok = xa >= 0 && xa + 3 < S;
        const int xc = xok ? xa : 0;
        rmask = 0;
#pragma unroll
        for (int u = 0; u < FN; ++u) {
            int rgi = 4 * u + wave;
            rgi = rgi < 105 ? rgi : 104;
            const int row = 3 * rgi + rig;
            const int p = row / SM_PH, py = row - p * SM_PH;
            const int ci = p / 3, dt = p - ci * 3;
            const int ti = 2 * to + dt - 1, yi = 4 * ty0 - 3 + py;
            if (xok && ti >= 0 && ti < T && yi >= 0 && yi < S) rmask |= 1u << u;
            const int tc = ti < 0 ? 0 : (ti >= T ? T - 1 : ti), yc = yi < 0 ? 0 : (yi >= S ? S - 1 : yi);
            v[u] = *reinterpret_cast<const f4a*>(clip + ((((int64_t)b * 3 + ci) * T + tc) * S + yc) * S + xc);
        }
    };
    auto fill_commit = [&](int tile, f4u (&v)[FN], uint32_t rmask, char* dst) {
#pragma unroll
        for (int u = 0; u < FN; ++u) {
            const int rgi = 4 * u + wave;
            const int row = 3 * rgi + rig;
            const float keep = ((rmask >> u) & 1) ? 1.f : 0.f;
            const float f0 = v[u][0] * keep, f1 = v[u][1] * keep, f2 = v[u][2] * keep, f3 = v[u][3] * keep;
            const float nx0 = __shfl_down(f0, 1, 64);          // first pixel of the next segment of the same row
            if (lane_on && seg < 17 && rgi < 105) {
                uint2 o;
                o.x = pack_bf16x2(f1, f2);
                o.y = pack_bf16x2(f3, nx0);
                *reinterpret_cast<uint2*>(dst + row * SM_ROWB + 8 * seg) = o;
            }
        }
    };

    // A-fragment lane base: token r of m-block mb -> (yo_l = 2*mb + (r>>4), xo_l = r&15)
    const int a_base = (4 * (r >> 4)) * SM_ROWB + 8 * (r & 15);
    const int hx = h * (SM_PLANE - 7 * SM_ROWB);   // extra offset when row 2s+1 starts the next (ci,dt) plane
    const float bias_v = (wave < 3) ? bias[32 * wave + r] : 0.f;

    int tile = blockIdx.x;
    f4u fv[FN];
    uint32_t fmask = 0;
    if (!(STEM_ABL & 1) && tile < ntiles) {
        fill_issue(tile, fv, fmask);
        fill_commit(tile, fv, fmask, patch[0]);
    }
    __syncthreads();
    int buf = 0;
    for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        const bool more = !(STEM_ABL & 1) && tile + (int)gridDim.x < ntiles;
        if (more) fill_issue(tile + gridDim.x, fv, fmask);      // lands under the MFMA work below
        if (wave < 3) {
            const char* pb = patch[buf] + a_base;
            f32x16 acc[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mb][i] = 0.f;
#pragma unroll
            for (int s = 0; s < ((STEM_ABL & 2) ? 1 : 32); ++s) {
                const int row0 = 2 * s;                       // tap row of lane-half 0
                const int c0 = (row0 / 7) * SM_PLANE + (row0 % 7) * SM_ROWB;
                int off = c0;
                if (s < 31) {
                    off += h * SM_ROWB;
                    if (row0 % 7 == 6) off += hx;
                }                                             // s == 31: row 63 has zero weights, read row 62 again
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    const char* ap = pb + off + mb * (8 * SM_ROWB);
                    const uint2 lo = *reinterpret_cast<const uint2*>(ap);
                    const uint2 hi = *reinterpret_cast<const uint2*>(ap + 8);
                    uint4 u = make_uint4(lo.x, lo.y, hi.x, hi.y);
                    acc[mb] = mfma16(*reinterpret_cast<bf16x8*>(&u), bfrag[s], acc[mb]);
                }
            }
            // epilogue: + bias + pos_spatial[hw] + pos_temporal[t], token-major fp32
            const int b = tile / (To * tiles_per_frame);
            int rem = tile - b * (To * tiles_per_frame);
            const int to = rem / tiles_per_frame;
            rem -= to * tiles_per_frame;
            const int ty0 = (rem / tiles_x) * SM_TY, tx0 = (rem % tiles_x) * SM_TX;
            const int c = 32 * wave + r;
            const float add_t = bias_v, pt = pos_t[to * 96 + c];
            // the 16 position-embedding loads of an m-block are all requested (at clamped token positions) before the first add:
            // a bounds test around each load made the compiler wait for every single one (64 round trips per tile, 4x the tile's MFMA time)
            float* xt = x + ((int64_t)b * To + to) * So * So * 96 + c;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                float ps[16];
                int hwv[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int tl = (i & 3) + 8 * (i >> 2) + 4 * h;      // token within the m-block
                    const int yo = ty0 + 2 * mb + (tl >> 4), xo = tx0 + (tl & 15);
                    const bool ok = yo < So && xo < So;
                    const int hw = ok ? yo * So + xo : 0;
                    hwv[i] = ok ? hw : -1;
                    ps[i] = pos_s[(int64_t)hw * 96 + c];
                }
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (hwv[i] >= ((STEM_ABL & 4) ? 1 << 30 : 0)) xt[(int64_t)hwv[i] * 96] = (acc[mb][i] + add_t) + (ps[i] + pt);
            }
        }
        if (more) fill_commit(tile + gridDim.x, fv, fmask, patch[buf ^ 1]);
        __syncthreads();
    }
}


// ------------------------------------------------------------------------------------------------
// v3 (round 4): the same implicit GEMM, every input frame requested ONCE per spatial tile.
//   A workgroup owns a contiguous run of the flattened (clip, spatial tile, output frame) order -- output frame innermost -- and keeps
//   a ring of 16-bit input frames in LDS: output frame `to` reads input frames 2to-1, 2to, 2to+1, so stepping to the next output frame of
//   the same spatial tile needs two new frames (6 planes of 35 x 68 pixels), not nine planes; frame -1 is a permanently zero slot.
//   n = B * tiles * To tile-frames are split evenly over the workgroups (6272 = 24.5 x 256 at the bench size: 98 % balanced, against
//   75 % for whole spatial tiles).
//   K ordering: tap row rho = (dt * 3 + ci) * 7 + dy (63 rows + 1 zero row), 8 columns per row: column j is image pixel 4 xo - 4 + j,
//   i.e. the zero weight sits in column 0 and the LDS image keeps the 16-byte ALIGNED quads of the clip as they are loaded (no shift).
//   Fill: wave 3 is a loader (see below): the two new frames of a tile-frame travel by LDS-DMA into an fp32 staging area one tile-frame
//   ahead and are converted once into the ring (16 B of fp32 -> 8 B of 16 bit per piece, 28 pieces per lane and frame at offsets computed
//   once per kernel); pieces outside the image request a valid dummy address and are replaced by zero when they are converted -- no
//   bounds test sits between a request and its use.
//   The fp32 weights reach the B fragments through LDS: converted once to 16 bit with coalesced 16-byte loads, then gathered with 2-byte
//   LDS reads, instead of 256 strided 4-byte global loads per lane.
//   Epilogue: the spatial position embedding of the tile's tokens stays in registers for the whole run over output frames.
// ------------------------------------------------------------------------------------------------
#define SR_PITCH 160                           // 68 pixels x 2 B = 136 -> 160: the two 16-lane halves of a fragment read hit disjoint banks
#define SR_ROWS 105                            // 3 ci x 35 patch rows
#define SR_FRAME (SR_ROWS * SR_PITCH)          // 16,800 B
#define SR_NSLOT 5                             // 3 frames in use + 2 being filled
#define SR_STAGE 100864                         // fp32 staging of the LDS-DMA behind the ring + zero frame (6 x 16,800 = 100,800)
#define SR_STGF (SR_FU * 1024)                 // one frame of staging: 1792 quads
#define SR_POST (SR_STAGE + 2 * SR_STGF)       // the temporal position embedding [To][96] fp32 (To <= SR_MAXTO)
#define SR_MAXTO 14
#define SR_LDS (SR_POST + SR_MAXTO * 384)      // 163,584 B of 163,840
#define SR_PIECES (SR_ROWS * 17)               // 16-byte quads per frame
#define SR_FU 28                               // pieces per loader lane and frame (64 lanes)

typedef uint32_t sr_u4 __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int sr_fo(int rho) { return (((rho % 21) / 7) * 35 + (rho % 7)) * SR_PITCH; }   // tap row -> offset inside its frame

// The matrix loop of a tile-frame: 128 (k-step s, token block mb) products, software-pipelined by hand -- left to the compiler every
// fragment read was issued right before its MFMA behind an lgkmcnt(0) (107 us of the kernel for 42 us of matrix work).  SR_D fragments
// are in flight, one ds_read2_b64 each (two 8-byte halves: a token's 8 pixels start on an 8-byte, not a 16-byte boundary); LDS returns
// in order, so waiting for all but the SR_D - 1 youngest reads releases the oldest fragment.
#define SR_D 8
struct SrBases { uint32_t fb2, sx, p1[3], px[3]; };

// product I of a tile-frame = (token block mb = I >> 5, k-step s = I & 31): token-block-major, so that the 16 stores of a finished
// block's accumulators can be issued between the MFMAs of the next block
template <int I> __device__ __forceinline__ void sr_read(bf16x8& f, const SrBases& bs) {
    constexpr int s = I & 31, mb = I >> 5;
    constexpr int r0 = 2 * s, r1 = s < 31 ? 2 * s + 1 : 62;
    constexpr bool last = r1 == r0, cross = !last && r0 / 21 != r1 / 21;
    constexpr int o = (cross ? 0 : sr_fo(r0)) + (mb >> 1) * 16 * SR_PITCH;
    const uint32_t a = (last ? bs.fb2 : cross ? bs.sx : (sr_fo(r1) - sr_fo(r0) == SR_PITCH ? bs.p1[r0 / 21] : bs.px[r0 / 21])) + o;
    asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(f) : "v"(a), "n"((mb & 1) * SR_PITCH), "n"((mb & 1) * SR_PITCH + 1));
}

// epilogue of one accumulator element: (conv + bias) + (pos_spatial[hw] + pos_temporal[to]), token-major fp32.  Element (mb, e) of lane
// half h is token row ty0 + 2 mb + (e >> 3), column tx0 + (e & 3) + 8 ((e >> 2) & 1) + 4 h; `xr` points at (ty0, tx0 + 4 h), channel c.
struct SrOut { float* x0; uint32_t row_bytes, voff[2]; int ylim, xlim; float bias_v, pt; };
// x0: token (ty0, tx0) of the output frame (wave-uniform); voff: this lane's byte offsets (token 4h resp. 4h + 8 of a row, channel c);
// ylim / xlim: rows / columns of the tile inside the frame (xlim counted from the lane's first token).
// The store is inline assembly so that it stays where it is written, between two MFMAs (the compiler gathered plain stores behind the
// matrix loop).  Consequence: the compiler's vmcnt bookkeeping does not see these stores -- any global LOAD the compute waves wait for
// also waits for every older store, which is why the temporal position embedding is read from LDS and the spatial one once per tile.
template <int MB, int E, bool PRED> __device__ __forceinline__ void sr_store(const SrOut& o, const f32x16 (&acc)[4], const float (&ps)[4][16]) {
    constexpr int row = 2 * MB + (E >> 3), col = 8 * ((E >> 2) & 1) + (E & 3);
    const float v = (acc[MB][E] + o.bias_v) + (ps[MB][E] + o.pt);
    const uint32_t off = o.voff[(E >> 2) & 1] + (uint32_t)row * o.row_bytes;
    if (!PRED || (row < o.ylim && col < o.xlim)) asm volatile("global_store_dword %0, %1, %2 offset:%3" ::"v"(off), "v"(v), "s"(o.x0), "n"((E & 3) * 384) : "memory");
}

template <int I, int N, bool PRED> __device__ __forceinline__ void sr_steps(bf16x8 (&f)[SR_D], f32x16 (&acc)[4], const bf16x8 (&bfrag)[32], const SrBases& bs,
                                                                          const SrOut& o, const float (&ps)[4][16]) {
    if constexpr (I < N) {
        constexpr int left = N - 1 - I < SR_D - 1 ? N - 1 - I : SR_D - 1;     // younger fragments in flight
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f[I % SR_D]) : "n"(left));
        acc[I >> 5] = mfma16(f[I % SR_D], bfrag[I & 31], acc[I >> 5]);
        if constexpr (I + SR_D < N) sr_read<I + SR_D>(f[I % SR_D], bs);
        if constexpr (I >= 32 && (I & 1) == 0) {
            constexpr int pmb = I / 32 - 1, pe = (I % 32) / 2;
            sr_store<pmb, pe, PRED>(o, acc, ps);
        }
        sr_steps<I + 1, N, PRED>(f, acc, bfrag, bs, o, ps);
    }
}

template <int E, bool PRED> __device__ __forceinline__ void sr_store_tail(const SrOut& o, const f32x16 (&acc)[4], const float (&ps)[4][16]) {
    if constexpr (E < 16) {
        sr_store<3, E, PRED>(o, acc, ps);
        sr_store_tail<E + 1, PRED>(o, acc, ps);
    }
}

template <int I> __device__ __forceinline__ void sr_prime(bf16x8 (&f)[SR_D], const SrBases& bs) {
    if constexpr (I < SR_D) {
        sr_read<I>(f[I], bs);
        sr_prime<I + 1>(f, bs);
    }
}

__global__ __launch_bounds__(256, 1) void stem_ring_kernel(const float* __restrict__ clip, const float* __restrict__ w,
                                                           const float* __restrict__ bias, const float* __restrict__ pos_s,
                                                           const float* __restrict__ pos_t, float* __restrict__ x, int B,
                                                           int T, int S, int To, int So, int tiles_x, int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int tpf = tiles_x * tiles_y;
    const int n = B * tpf * To;
    const int G = gridDim.x;
    int g = blockIdx.x;
    if ((G & 7) == 0) g = (g & 7) * (G >> 3) + (g >> 3);          // neighbouring runs (shared halos) on one XCD's L2
    const int per = n / G, rem = n - per * G;
    const int start = g * per + (g < rem ? g : rem);
    const int end = start + per + (g < rem ? 1 : 0);
    if (start >= end) return;

    // ---- B fragments: W[n = 32 wave + r][rho = 2s + h][column j].  All 96 x 441 weights go through LDS as 16-bit values (coalesced 16-byte
    //      loads, converted once), then every compute lane gathers its 32 fragments with 2-byte LDS reads. ---------------------------------------
    bf16x8 bfrag[32];
    {
        constexpr int NQ = 96 * 441 / 4;                           // float4 pieces
#pragma unroll 1
        for (int k = 0; k < 3; ++k) {
            float4 v[14];
#pragma unroll
            for (int j = 0; j < 14; ++j) {
                const int idx = tid + 256 * (14 * k + j);
                v[j] = reinterpret_cast<const float4*>(w)[idx < NQ ? idx : NQ - 1];
            }
#pragma unroll
            for (int j = 0; j < 14; ++j) {
                const int idx = tid + 256 * (14 * k + j);
                if (idx < NQ) reinterpret_cast<uint2*>(smem)[idx] = make_uint2(pack_bf16x2(v[j].x, v[j].y), pack_bf16x2(v[j].z, v[j].w));
            }
        }
        __syncthreads();
        if (wave < 3) {
            const uint16_t* wl = reinterpret_cast<const uint16_t*>(smem) + (32 * wave + r) * 441;
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                const int rho = 2 * s + h;
                const int rc = rho < 63 ? rho : 62;
                const int dt = rc / 21, rm = rc - 21 * dt, ci = rm / 7, dy = rm - 7 * ci;
                const uint16_t* wr = wl + ((ci * 3 + dt) * 7 + dy) * 7;
                uint32_t f[8];
                f[0] = 0u;
#pragma unroll
                for (int j = 1; j < 8; ++j) f[j] = rho < 63 ? (uint32_t)wr[j - 1] : 0u;
                const uint4 u = make_uint4(f[0] | (f[1] << 16), f[2] | (f[3] << 16), f[4] | (f[5] << 16), f[6] | (f[7] << 16));
                bfrag[s] = *reinterpret_cast<const bf16x8*>(&u);
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < SR_FRAME / 16; i += 256) reinterpret_cast<uint4*>(smem + SR_NSLOT * SR_FRAME)[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < To * 96; i += 256) reinterpret_cast<float*>(smem + SR_POST)[i] = pos_t[i];

    // ---- wave 3 is the loader.  The two new frames of tile-frame i + 2 travel by LDS-DMA (global_load_lds_dwordx4: 28 + 28 pieces of 16 B
    //      per lane, no register round trip) into an fp32 staging area while the frames of tile-frame i + 1, which landed during the previous
    //      step, are converted to 16 bit and written into the two free ring slots; waves 0-2 (one 32-channel block each) run the matrix
    //      work and the epilogue of tile-frame i meanwhile.  A lane converts exactly the pieces it requested (the DMA destination is
    //      lane-linear), so its own vmcnt orders staging reads behind the copies; one barrier per tile-frame publishes the ring. ------------
    const int64_t clip_stride = (int64_t)3 * T * S * S;            // floats per clip
    auto decode = [&](int i, int& b, int& sp, int& to) {
        b = i / (tpf * To);
        const int rm = i - b * (tpf * To);
        sp = rm / To;
        to = rm - sp * To;
    };
    int b, sp, to;
    decode(start, b, sp, to);
    int q = to > 0 ? 3 : 2;                                        // ring position of the next frame to land (mod SR_NSLOT)

    if (wave == 3) {
        // piece p = lane + 64 u -> (ci, patch row, quad)
        int goff[SR_FU], loff[SR_FU];
#pragma unroll
        for (int u = 0; u < SR_FU; ++u) {
            const int p = lane + 64 * u;
            const int pc = p < SR_PIECES ? p : SR_PIECES - 1;
            const int row = pc / 17, seg = pc - 17 * row;
            const int ci = row / 35, py = row - 35 * ci;
            goff[u] = ((ci * T * S + py) * S + 4 * seg) * 4;
            loff[u] = row * SR_PITCH + 8 * seg;
        }
        const uint32_t stg = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)(smem) + SR_STAGE;
        auto keep_mask = [&](int ty0, int tx0) {
            uint32_t m = 0;
#pragma unroll
            for (int u = 0; u < SR_FU; ++u) {
                const int p = lane + 64 * u;
                const int pc = p < SR_PIECES ? p : SR_PIECES - 1;
                const int row = pc / 17, seg = pc - 17 * row;
                const int yi = 4 * ty0 - 3 + row % 35, xa = 4 * tx0 - 4 + 4 * seg;
                if (yi >= 0 && yi < S && xa >= 0 && xa < S) m |= 1u << u;
            }
            return m;
        };
        // A tile whose 35 x 68 patch lies inside the image needs no replacement at all (wave-uniform test; 1 << 31 in the mask).
        auto tile_mask = [&](int ty0, int tx0) {
            const bool inside = ty0 > 0 && tx0 > 0 && 4 * ty0 + 31 < S && 4 * tx0 + 63 < S;
            return inside ? 0x80000000u : keep_mask(ty0, tx0);
        };
        // frame ti of clip bb, patch origin of tile (ty0, tx0) -> staging half k.  Pieces outside the image request the clip's first
        // quad instead (any valid address) and are replaced by zero when they are converted.
        auto dma_frame = [&](int k, int bb, int ty0, int tx0, int ti, uint32_t mask) {
            const float* base = clip + bb * clip_stride;
            const int toff = ((ti * S + 4 * ty0 - 3) * S + 4 * tx0 - 4) * 4;
            const bool inside = __builtin_amdgcn_readfirstlane(mask) >> 31;
#pragma unroll
            for (int u = 0; u < SR_FU; ++u) {
                uint32_t off = (uint32_t)(goff[u] + toff);
                if (!inside) off = ((mask >> u) & 1) ? off : 0u;
                const uint32_t m0v = __builtin_amdgcn_readfirstlane(stg + (uint32_t)(k * SR_STGF + u * 1024));
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(off), "s"(base) : "memory");
            }
        };
        auto convert = [&](int k, uint32_t mask, int slot) {
            const char* src = smem + SR_STAGE + k * SR_STGF + 16 * lane;
            char* dst = smem + slot * SR_FRAME;
            const bool inside = __builtin_amdgcn_readfirstlane(mask) >> 31;
            const bool last_on = lane < SR_PIECES - 64 * (SR_FU - 1);
#pragma unroll
            for (int u = 0; u < SR_FU; ++u) {
                const float4 v = *reinterpret_cast<const float4*>(src + u * 1024);
                uint2 o = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
                if (!inside) {
                    const bool keep = (mask >> u) & 1;
                    o.x = keep ? o.x : 0u;
                    o.y = keep ? o.y : 0u;
                }
                if (u < SR_FU - 1 || last_on) *reinterpret_cast<uint2*>(dst + loff[u]) = o;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the staging reads have returned: the half may be requested again
        };
        auto landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
        int ty0 = (sp / tiles_x) * SM_TY, tx0 = (sp % tiles_x) * SM_TX;
        uint32_t cmask = tile_mask(ty0, tx0);                      // of the tile-frame whose frames are converted next
        {
            int qq = 0;
            if (to > 0) {
                dma_frame(0, b, ty0, tx0, 2 * to - 1, cmask);
                landed();
                convert(0, cmask, 0);
                qq = 1;
            }
            dma_frame(0, b, ty0, tx0, 2 * to, cmask);
            dma_frame(1, b, ty0, tx0, 2 * to + 1, cmask);
            landed();
            convert(0, cmask, qq);
            convert(1, cmask, qq + 1);
        }
        // The request stream runs two tile-frames ahead of the matrix work, one ahead of the conversions: half k of the staging area is
        // requested again as soon as it has been converted.  (Tried: a second request channel through 2 x 28 quads of this wave's
        // accumulator registers to give every copy two tile-frames to land -- the register allocator spilled the loader, left out.)
        int rb = b, rsp = sp, rto = to, rty0 = ty0, rtx0 = tx0;
        uint32_t rmask = cmask;
        auto request_pos = [&](int i) {
            int nb, nsp;
            decode(i, nb, nsp, rto);
            if (nsp != rsp || nb != rb) {
                rty0 = (nsp / tiles_x) * SM_TY;
                rtx0 = (nsp % tiles_x) * SM_TX;
                rmask = tile_mask(rty0, rtx0);
                rb = nb; rsp = nsp;
            }
        };
        uint32_t nmask = cmask;                                    // mask of the frames in flight
        if (!(STEM_ABL & 1) && start + 1 < end) {
            request_pos(start + 1);
            dma_frame(0, rb, rty0, rtx0, 2 * rto, rmask);
            dma_frame(1, rb, rty0, rtx0, 2 * rto + 1, rmask);
            nmask = rmask;
        }
        __syncthreads();
        for (int i = start; i < end; ++i) {
            if (!(STEM_ABL & 1) && i + 1 < end) {
                const bool again = i + 2 < end;
                const uint32_t lmask = nmask;
                if (again) request_pos(i + 2);
                landed();
                convert(0, lmask, q);
                if (again) dma_frame(0, rb, rty0, rtx0, 2 * rto, rmask);          // half 0 is free again: request it before converting half 1
                convert(1, lmask, q + 1 < SR_NSLOT ? q + 1 : q + 1 - SR_NSLOT);
                if (again) dma_frame(1, rb, rty0, rtx0, 2 * rto + 1, rmask);
                q = q + 2 < SR_NSLOT ? q + 2 : q + 2 - SR_NSLOT;
                nmask = rmask;
            }
            __syncthreads();
        }
        return;
    }

    // ---- waves 0-2: matrix work + epilogue -------------------------------------------------------------------------------------------------
    __syncthreads();                                               // the first tile-frame's frames are in the ring
    const int c = 32 * wave + r;
    const float bias_v = bias[c];
    const uint32_t lane_off = (uint32_t)((4 * (r >> 4)) * SR_PITCH + 8 * (r & 15));
    const uint32_t lane_c = (uint32_t)(4 * h * 96 + c);           // token 4h of a row group, channel c
    float ps[4][16];
    int ps_sp = -1;
    for (int i = start; i < end; ++i) {
        const int ty0 = (sp / tiles_x) * SM_TY, tx0 = (sp % tiles_x) * SM_TX;
        {
            // frames of this tile: ring positions q-3, q-2, q-1 (dt = 0, 1, 2); the first output frame reads the zero frame for dt = 0
            const int s2 = (q + SR_NSLOT - 1) % SR_NSLOT, s1 = (q + SR_NSLOT - 2) % SR_NSLOT;
            const int s0 = to == 0 ? SR_NSLOT : (q + SR_NSLOT - 3) % SR_NSLOT;
            const uint32_t fb[3] = {(uint32_t)(s0 * SR_FRAME) + lane_off, (uint32_t)(s1 * SR_FRAME) + lane_off, (uint32_t)(s2 * SR_FRAME) + lane_off};
            SrBases bs;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                bs.p1[d] = fb[d] + (uint32_t)(h * SR_PITCH);       // lane half 1 reads the next patch row ...
                bs.px[d] = fb[d] + (uint32_t)(h * 29 * SR_PITCH);  // ... or the first row of the next channel plane
            }
            bs.sx = h ? fb[1] + sr_fo(21) : fb[0] + sr_fo(20);     // tap rows 20 | 21 sit in different frames
            bs.fb2 = fb[2];                                        // s = 31: tap row 63 has zero weights, both halves read row 62
            if (ps_sp != sp) {                                     // the spatial position embedding of the tile's tokens: once per spatial tile
                ps_sp = sp;
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int yo = ty0 + 2 * mb + (e >> 3), xo = tx0 + (e & 3) + 8 * ((e >> 2) & 1) + 4 * h;
                        ps[mb][e] = pos_s[(int64_t)(yo < So && xo < So ? yo * So + xo : 0) * 96 + c];
                    }
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)                     // all 64 values have arrived before the matrix loop starts: no vmcnt wait inside it
#pragma unroll
                    for (int e = 0; e < 16; ++e) asm volatile("" ::"v"(ps[mb][e]));
            }
            SrOut o;
            o.x0 = x + (((int64_t)b * To + to) * So * So + ty0 * So + tx0) * 96;
            o.row_bytes = (uint32_t)So * 384u;
            o.voff[0] = lane_c * 4u;
            o.voff[1] = lane_c * 4u + 8u * 384u;
            o.ylim = (STEM_ABL & 4) ? 0 : So - ty0;
            o.xlim = So - tx0 - 4 * h;
            o.bias_v = bias_v;
            o.pt = reinterpret_cast<const float*>(smem + SR_POST)[to * 96 + c];
            f32x16 acc[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mb][e] = 0.f;
            bf16x8 fr[SR_D];
            sr_prime<0>(fr, bs);
            asm volatile("" ::"v"(o.pt));                            // (the LDS read of pt is waited for here, not inside the loop)
            // a token block's stores go out between the MFMAs of the next block; the last block's follow the loop
            constexpr int NP = (STEM_ABL & 2) ? 8 : 128;
            if (!(STEM_ABL & 4) && ty0 + SM_TY <= So && tx0 + SM_TX <= So) {
                sr_steps<0, NP, false>(fr, acc, bfrag, bs, o, ps);
                sr_store_tail<0, false>(o, acc, ps);
            } else {                                               // a tile that crosses the frame's edge: every store tests its token
                sr_steps<0, NP, true>(fr, acc, bfrag, bs, o, ps);
                sr_store_tail<0, true>(o, acc, ps);
            }
        }
        if (!(STEM_ABL & 1) && i + 1 < end) q = q + 2 < SR_NSLOT ? q + 2 : q + 2 - SR_NSLOT;
        __syncthreads();
        if (i + 1 < end) decode(i + 1, b, sp, to);
    }
}

extern "C" int mvit_stem_fwd(const float* clip, const float* w, const float* bias, const float* pos_spatial,
                             const float* pos_temporal, float* x, int B, int T, int S, int act_dtype, void* stream) {
    if (!clip || !w || !bias || !pos_spatial || !pos_temporal || !x || B <= 0 || T <= 0 || S <= 0) return MVIT_EINVAL;
    if ((T & 1) || (S & 3)) return MVIT_EUNSUPPORTED;
    const int To = T / 2, So = S / 4;
    if (act_dtype == MVIT_BF16) {
        const int tiles_x = (So + SM_TX - 1) / SM_TX, tiles_y = (So + SM_TY - 1) / SM_TY;
        const int ntiles = B * To * tiles_x * tiles_y;
        const int grid = ntiles < 256 ? ntiles : 256;
        static const bool v2 = getenv("MVIT_STEM_V2") && getenv("MVIT_STEM_V2")[0] == '1';        // A/B switch: the r2 kernel
        if (v2 || To > SR_MAXTO || (int64_t)3 * T * S * S * 4 >= ((int64_t)1 << 31)) {
            hipLaunchKernelGGL(stem_mfma_kernel, dim3(grid), dim3(256), 0, as_stream(stream), clip, w, bias, pos_spatial,
                               pos_temporal, x, B, T, S, To, So, tiles_x, tiles_y);
            MVIT_LAUNCH_CHECK();
            return MVIT_OK;
        }
        static DevFlags attr;
        DevFlag done = dev_flag(attr);
        if (!done) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_ring_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SR_LDS) != hipSuccess)
                return MVIT_ELAUNCH;
            done = true;
        }
        hipLaunchKernelGGL(stem_ring_kernel, dim3(grid), dim3(256), SR_LDS, as_stream(stream), clip, w, bias, pos_spatial,
                           pos_temporal, x, B, T, S, To, So, tiles_x, tiles_y);
        MVIT_LAUNCH_CHECK();
        return MVIT_OK;
    }
    const int tiles = ((So + ST_TX - 1) / ST_TX) * ((So + ST_TY - 1) / ST_TY);
    dim3 grid(tiles, To, B);
    hipLaunchKernelGGL(stem_f32_kernel, grid, dim3(256), 0, as_stream(stream), clip, w, bias, pos_spatial, pos_temporal,
                       x, T, S, To, So);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}
