// Weight gradients of the Linear layers:  dW[N][K] += sum_m dy[m][n] * a[m][k]   (a "TN" GEMM whose reduction
// dimension is the huge token dimension M).
//   * bf16 path: MFMA 32x32x16; one workgroup = a 96x96 tile of dW over a chunk of M.  Both operands have M
//     as the MFMA k dimension, so both fragments are TRANSPOSED reads (ds_read_b64_tr_b16) of row-major
//     [64 m][96] bf16 slabs; the four waves take one 16-row k-step of every slab each (9 MFMAs/wave/slab) and
//     write their partial tiles to the M chunk's slab (128 B contiguous per half-wave), summed in chunk order afterwards.
//   * fp32 path: exact VALU kernel (parity path).
// dy may carry a per-sample drop-path factor (row_scale), applied while staging.
#include <stdlib.h>

#include "common.h"

typedef __attribute__((address_space(3))) bf16x4 lds_b4;

#define WG_MCHUNK 1024
#define WG_ROWB 192

template <typename T>
__device__ __forceinline__ uint4 stage8(const T* p, float sc);
template <>
__device__ __forceinline__ uint4 stage8<bf16_t>(const bf16_t* p, float sc) {
    if (sc == 1.0f) return *reinterpret_cast<const uint4*>(p);
    float4 lo, hi;
    load8(p, lo, hi);
    uint4 r;
    r.x = pack_bf16x2(lo.x * sc, lo.y * sc); r.y = pack_bf16x2(lo.z * sc, lo.w * sc);
    r.z = pack_bf16x2(hi.x * sc, hi.y * sc); r.w = pack_bf16x2(hi.z * sc, hi.w * sc);
    return r;
}
template <>
__device__ __forceinline__ uint4 stage8<float>(const float* p, float sc) {
    float4 lo, hi;
    load8(p, lo, hi);
    uint4 r;
    r.x = pack_bf16x2(lo.x * sc, lo.y * sc); r.y = pack_bf16x2(lo.z * sc, lo.w * sc);
    r.z = pack_bf16x2(hi.x * sc, hi.y * sc); r.w = pack_bf16x2(hi.z * sc, hi.w * sc);
    return r;
}

template <typename TA, typename TD>
__global__ __launch_bounds__(192) void wgrad_mfma_kernel(const TA* __restrict__ a, int64_t lda, const TD* __restrict__ dy,
                                                         int64_t ldd, const float* __restrict__ row_scale, int64_t rps,
                                                         float* __restrict__ dW, float* __restrict__ db, int64_t M, int N, int K,
                                                         int mchunk, float* __restrict__ part) {
    // 3 waves; wave w owns the 32(n) x 96(k) strip n-block w of the 96x96 tile (48 accumulator registers -> 3-4
    // workgroups per CU) and walks all four 16-row k-steps of every 64-row slab; no cross-wave reduction.
    __shared__ __attribute__((aligned(16))) char smem[2 * 64 * WG_ROWB];
    char* sD = smem;                 // dy slab [64 m][96 n]
    char* sA = smem + 64 * WG_ROWB;  // a  slab [64 m][96 k]
    const int ntk = K / 96;
    const int n0 = (blockIdx.x / ntk) * 96, k0 = (blockIdx.x % ntk) * 96;
    const int64_t mbeg = (int64_t)blockIdx.y * mchunk;
    const int64_t mend = mbeg + mchunk < M ? mbeg + mchunk : M;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const bool do_bias = db != nullptr && k0 == 0;   // block-uniform
    f32x16 acc[3], bacc;
#pragma unroll
    for (int i = 0; i < 16; ++i) bacc[i] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 3; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[kb][i] = 0.f;
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = MVIT_ONE16;

    // staging: 64 rows x 12 chunks = 768 chunks per slab, 4 per thread
    int s_off[4];
    int s_row[4], s_chk[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 192 * i;
        s_row[i] = c / 12;
        s_chk[i] = c - s_row[i] * 12;
        s_off[i] = s_row[i] * WG_ROWB + s_chk[i] * 16;
    }
    // transposed fragment reads: rows 8h + {0..3 | 4..7} of a 16-row k-step; lane supplies row (i16>>2), cols 16*(gi&1)+4*(i16&3)
    const int i16 = lane & 15, gi = lane >> 4;
    const int t_off = (8 * h + (i16 >> 2)) * WG_ROWB + (16 * (gi & 1) + 4 * (i16 & 3)) * 2;

    uint4 rd[4], ra[4];
    auto gload = [&](int64_t m0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t m = m0 + s_row[i];
            if (m < mend) {
                const float sc = row_scale ? row_scale[m / rps] : 1.0f;
                rd[i] = stage8<TD>(dy + m * ldd + n0 + 8 * s_chk[i], sc);
                ra[i] = stage8<TA>(a + m * lda + k0 + 8 * s_chk[i], 1.0f);
            } else {
                rd[i] = make_uint4(0, 0, 0, 0);
                ra[i] = make_uint4(0, 0, 0, 0);
            }
        }
    };
    gload(mbeg);
    for (int64_t m0 = mbeg; m0 < mend; m0 += 64) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<uint4*>(sD + s_off[i]) = rd[i];
            *reinterpret_cast<uint4*>(sA + s_off[i]) = ra[i];
        }
        __syncthreads();
        if (m0 + 64 < mend) gload(m0 + 64);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const char* dp = sD + t_off + ks * 16 * WG_ROWB + wave * 64;
            const bf16x4 dlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b4*)(dp));
            const bf16x4 dhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b4*)(dp + 4 * WG_ROWB));
            bf16x8 df;
            df[0] = dlo[0]; df[1] = dlo[1]; df[2] = dlo[2]; df[3] = dlo[3];
            df[4] = dhi[0]; df[5] = dhi[1]; df[6] = dhi[2]; df[7] = dhi[3];
#pragma unroll
            for (int kb = 0; kb < 3; ++kb) {
                const char* ap = sA + t_off + ks * 16 * WG_ROWB + kb * 64;
                const bf16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b4*)(ap));
                const bf16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b4*)(ap + 4 * WG_ROWB));
                bf16x8 af;
                af[0] = alo[0]; af[1] = alo[1]; af[2] = alo[2]; af[3] = alo[3];
                af[4] = ahi[0]; af[5] = ahi[1]; af[6] = ahi[2]; af[7] = ahi[3];
                acc[kb] = mfma16(df, af, acc[kb]);
            }
            if (do_bias) bacc = mfma16(df, ones, bacc);   // dy^T . 1 = bias gradient
        }
    }
    // acc[kb][i]: row n = 32*wave + (i&3) + 8(i>>2) + 4h, col k = 32kb + r  -> 128 contiguous bytes per half-wave
    // this M chunk's tile goes to its own slab (plain stores; wgrad_reduce_kernel adds the slabs in chunk order: bit-reproducible)
    {
        float* oW = part + (int64_t)blockIdx.y * ((int64_t)N * K + N);
#pragma unroll
        for (int kb = 0; kb < 3; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int n = n0 + 32 * wave + (i & 3) + 8 * (i >> 2) + 4 * h;
                oW[(int64_t)n * K + k0 + 32 * kb + r] = acc[kb][i];
            }
        if (do_bias && r == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) oW[(int64_t)N * K + n0 + 32 * wave + (i & 3) + 8 * (i >> 2) + 4 * h] = bacc[i];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Large-tile weight gradient: dW[n0:+128][k0:+192] += sum over an M chunk of dy[m][n]^T a[m][k] (both operands bf16,
// row-major over m, so the contraction index is the ROW index of both).  Same skeleton as linear_big_kernel:
// 4 waves (2 x 2, 64n x 96k each), 64-row slabs (dy: 256-B row pieces, a: 384-B) double-buffered in LDS by
// global_load_lds_dwordx4, fragments by ds_read_b64_tr_b16 (transposing read: 8 consecutive m of one column), issued as
// inline asm two k-steps ahead of the MFMAs with counted lgkmcnt waits.
// LDS images: 64-byte segments of a row are XOR-permuted so that the four rows a transposing read touches fall in
// different bank ranges: dy image (256-B rows) seg ^= row&3, a image (384-B rows) seg ^= (row>>1)&1; the DMA applies
// the permutation to the SOURCE address, the reads to the LDS address.
// Partial tiles of the M chunks go to per-chunk slabs summed in chunk order (dW zeroed by the caller), db through a ones-MFMA.
// ------------------------------------------------------------------------------------------------
#define WB_BP 128
#define WB_BQ 192
#define WB_PROWB 256
#define WB_QROWB 384
#define WB_PANEL_P (64 * WB_PROWB)    // 16384
#define WB_PANEL_Q (64 * WB_QROWB)    // 24576
#define WB_BUF (WB_PANEL_P + WB_PANEL_Q)
#define WB_SMEM (2 * WB_BUF)

typedef __attribute__((address_space(1))) const void wb_gptr_t;
typedef __attribute__((address_space(3))) void wb_lptr_t;

template <int OFF>
__device__ __forceinline__ bf16x4 wb_tr(uint32_t addr) {
    bf16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
#define wb_wait5(F, N) \
    asm volatile("s_waitcnt lgkmcnt(%10)" \
                 : "+v"(F[0][0]), "+v"(F[0][1]), "+v"(F[1][0]), "+v"(F[1][1]), "+v"(F[2][0]), "+v"(F[2][1]), "+v"(F[3][0]), "+v"(F[3][1]), \
                   "+v"(F[4][0]), "+v"(F[4][1]) \
                 : "n"(N))
__device__ __forceinline__ bf16x8 wb_join(const bf16x4 (&f)[2]) {
    bf16x8 v;
    v[0] = f[0][0]; v[1] = f[0][1]; v[2] = f[0][2]; v[3] = f[0][3];
    v[4] = f[1][0]; v[5] = f[1][1]; v[6] = f[1][2]; v[7] = f[1][3];
    return v;
}

// KHALF (K <= 96: the 96-wide layers of blocks 0 and 1, 0.9 ms of weight gradients per step): a 192-wide k tile would be half
// padding -- the wq = 1 waves multiplied clamped columns whose results were dropped.  Here the a image is 64 rows x 192 B (three
// DMA pieces per wave instead of six, no swizzle needed: consecutive 192-byte rows already fall in different bank quarters), both
// wave columns work on the same 96 k columns and split the 64 slab rows between them (wq = 0: rows 0-31, wq = 1: rows 32-63:
// two k-steps each), and the two partial tiles meet in LDS after the last slab (wq = 1 parks, wq = 0 adds and stores).
template <bool KHALF>
__global__ __launch_bounds__(256, 2) void wgrad_big_kernel(const bf16_t* __restrict__ a, int64_t lda,
                                                           const bf16_t* __restrict__ dy, int64_t ldd,
                                                           float* __restrict__ dW, float* __restrict__ db, int64_t M, int N,
                                                           int K, int mchunk, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int QROWB = KHALF ? 192 : WB_QROWB, PANEL_Q = 64 * QROWB, BUF = WB_PANEL_P + PANEL_Q, NQ = KHALF ? 3 : 6, CPR = QROWB / 16;
    const int ntq = KHALF ? 1 : (K + WB_BQ - 1) / WB_BQ;      // ragged last tiles (N, K multiples of 32): sources clamped, results masked
    const int n0 = (blockIdx.x / ntq) * WB_BP, k0 = (blockIdx.x % ntq) * WB_BQ;
    const int64_t mbeg = (int64_t)blockIdx.y * mchunk;
    const int64_t mend = mbeg + mchunk < M ? mbeg + mchunk : M;     // multiple of 64 (checked by the launcher)
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wp = wave >> 1, wq = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const bool do_bias = db != nullptr && k0 == 0 && (KHALF || wq == 0);       // wave-uniform

    // DMA sources (per lane, relative to the slab's first row)
    int64_t p_src[4], q_src[NQ];
#pragma unroll
    for (int i = 0; i < 4; ++i) {       // dy image: piece = 4 rows x 256 B
        const int row = 4 * (4 * wave + i) + (lane >> 4), c = lane & 15;
        const int seg = (c >> 2) ^ (row & 3);
        int col = n0 + 8 * (4 * seg + (c & 3));
        col = col + 8 <= N ? col : N - 8;                  // columns past N: re-read valid memory (those accumulators are dropped)
        p_src[i] = (int64_t)row * ldd + col;
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {      // a image: 24 (12) chunks per row, pieces run across rows
        const int x = 64 * (NQ * wave + i) + lane;
        const int row = x / CPR, c = x - row * CPR;
        const int seg = KHALF ? (c >> 2) : ((c >> 2) ^ ((row >> 1) & 1));
        int col = k0 + 8 * (4 * seg + (c & 3));
        col = col + 8 <= K ? col : K - 8;
        q_src[i] = (int64_t)row * lda + col;
    }
    auto dma = [&](int64_t m0, int buf) {
        char* base = smem + buf * BUF;
        const bf16_t* ps = dy + m0 * ldd;
        const bf16_t* qs = a + m0 * lda;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((wb_gptr_t*)(ps + p_src[i]), (wb_lptr_t*)(base + 1024 * (4 * wave + i)), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < NQ; ++i)
            __builtin_amdgcn_global_load_lds((wb_gptr_t*)(qs + q_src[i]), (wb_lptr_t*)(base + WB_PANEL_P + 1024 * (NQ * wave + i)), 16, 0, 0);
    };

    // transposing fragment reads: lane supplies row 8h + (i16>>2) (+4 for the second half) and 8 bytes at
    // 32*(gi&1) + 8*(i16&3) inside the 64-byte segment of its 32-column block
    const int i16 = lane & 15, gi = lane >> 4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int rr = 8 * h + (i16 >> 2), inseg = 32 * (gi & 1) + 8 * (i16 & 3);
    uint32_t pa[2], qa[3];
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) pa[pb] = lds0 + (rr + (KHALF ? 32 * wq : 0)) * WB_PROWB + 64 * ((2 * wp + pb) ^ (i16 >> 2)) + inseg;
#pragma unroll
    for (int qb = 0; qb < 3; ++qb)
        qa[qb] = KHALF ? lds0 + WB_PANEL_P + (rr + 32 * wq) * QROWB + 64 * qb + inseg
                       : lds0 + WB_PANEL_P + rr * QROWB + 64 * ((3 * wq + qb) ^ ((i16 >> 3) & 1)) + inseg;

    f32x16 acc[2][3], bacc[2];
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
#pragma unroll
        for (int i = 0; i < 16; ++i) bacc[pb][i] = 0.f;
#pragma unroll
        for (int qb = 0; qb < 3; ++qb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[pb][qb][i] = 0.f;
    }
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = MVIT_ONE16;

    const int nslab = (int)((mend - mbeg) / 64);
    dma(mbeg, 0);
    for (int sl = 0; sl < nslab; ++sl) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (sl + 1 < nslab) dma(mbeg + 64 * (int64_t)(sl + 1), (sl + 1) & 1);
        const uint32_t bo = (sl & 1) ? BUF : 0;
        bf16x4 f[KHALF ? 2 : 4][5][2];      // [k-step][P0 P1 Q0 Q1 Q2][lo hi]
#define RD(KS) { \
            f[KS][0][0] = wb_tr<KS * 16 * WB_PROWB>(pa[0] + bo); f[KS][0][1] = wb_tr<KS * 16 * WB_PROWB + 4 * WB_PROWB>(pa[0] + bo); \
            f[KS][1][0] = wb_tr<KS * 16 * WB_PROWB>(pa[1] + bo); f[KS][1][1] = wb_tr<KS * 16 * WB_PROWB + 4 * WB_PROWB>(pa[1] + bo); \
            f[KS][2][0] = wb_tr<KS * 16 * QROWB>(qa[0] + bo); f[KS][2][1] = wb_tr<KS * 16 * QROWB + 4 * QROWB>(qa[0] + bo); \
            f[KS][3][0] = wb_tr<KS * 16 * QROWB>(qa[1] + bo); f[KS][3][1] = wb_tr<KS * 16 * QROWB + 4 * QROWB>(qa[1] + bo); \
            f[KS][4][0] = wb_tr<KS * 16 * QROWB>(qa[2] + bo); f[KS][4][1] = wb_tr<KS * 16 * QROWB + 4 * QROWB>(qa[2] + bo); }
#define MM(KS) { \
            const bf16x8 p0 = wb_join(f[KS][0]), p1 = wb_join(f[KS][1]); \
            const bf16x8 q0 = wb_join(f[KS][2]), q1 = wb_join(f[KS][3]), q2 = wb_join(f[KS][4]); \
            acc[0][0] = mfma16(p0, q0, acc[0][0]); acc[1][0] = mfma16(p1, q0, acc[1][0]); \
            acc[0][1] = mfma16(p0, q1, acc[0][1]); acc[1][1] = mfma16(p1, q1, acc[1][1]); \
            acc[0][2] = mfma16(p0, q2, acc[0][2]); acc[1][2] = mfma16(p1, q2, acc[1][2]); \
            if (do_bias) { bacc[0] = mfma16(p0, ones, bacc[0]); bacc[1] = mfma16(p1, ones, bacc[1]); } }
        RD(0) RD(1)
        wb_wait5(f[0], 10);
        if constexpr (KHALF) {
            MM(0)
            __builtin_amdgcn_sched_barrier(0);
            wb_wait5(f[1], 0);
            MM(1)
        } else {
            RD(2)
            MM(0)
            __builtin_amdgcn_sched_barrier(0);
            wb_wait5(f[1], 10);
            RD(3)
            MM(1)
            __builtin_amdgcn_sched_barrier(0);
            wb_wait5(f[2], 10);
            MM(2)
            __builtin_amdgcn_sched_barrier(0);
            wb_wait5(f[3], 0);
            MM(3)
        }
#undef RD
#undef MM
    }
    if constexpr (KHALF) {      // the two row halves meet: wq = 1 parks its tile (lane-contiguous), wq = 0 adds it
        float* park = reinterpret_cast<float*>(smem);
        __syncthreads();        // every wave is done reading the last slab
        if (wq == 1) {
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
#pragma unroll
                for (int qb = 0; qb < 3; ++qb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) park[((((wp * 2 + pb) * 3 + qb) * 16 + i) << 6) + lane] = acc[pb][qb][i];
#pragma unroll
                for (int i = 0; i < 16; ++i) park[12288 + (((wp * 2 + pb) * 16 + i) << 6) + lane] = bacc[pb][i];
            }
        }
        __syncthreads();
        if (wq == 1) return;
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
#pragma unroll
            for (int qb = 0; qb < 3; ++qb)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[pb][qb][i] += park[((((wp * 2 + pb) * 3 + qb) * 16 + i) << 6) + lane];
#pragma unroll
            for (int i = 0; i < 16; ++i) bacc[pb][i] += park[12288 + (((wp * 2 + pb) * 16 + i) << 6) + lane];
        }
    }
    // acc[pb][qb][i]: row n = n0 + 64wp + 32pb + (i&3) + 8(i>>2) + 4h, col k = k0 + 96wq + 32qb + r -> 128 contiguous bytes
    {      // this M chunk's tile to its own slab, plain stores (summed in chunk order by wgrad_reduce_kernel)
        float* oW = part + (int64_t)blockIdx.y * ((int64_t)N * K + N);
#pragma unroll
        for (int pb = 0; pb < 2; ++pb)
#pragma unroll
            for (int qb = 0; qb < 3; ++qb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int n = n0 + 64 * wp + 32 * pb + (i & 3) + 8 * (i >> 2) + 4 * h;
                    const int kk = k0 + (KHALF ? 0 : 96 * wq) + 32 * qb + r;
                    if (n < N && kk < K) oW[(int64_t)n * K + kk] = acc[pb][qb][i];
                }
        if (do_bias && r == 0) {
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int n = n0 + 64 * wp + 32 * pb + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (n < N) oW[(int64_t)N * K + n] = bacc[pb][i];
                }
        }
    }
}

// exact fp32: 64(n) x 64(k) tile per block over an M chunk, 4x4 per thread
__global__ __launch_bounds__(256) void wgrad_f32_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ dy,
                                                        int64_t ldd, const float* __restrict__ row_scale, int64_t rps,
                                                        float* __restrict__ dW, float* __restrict__ db, int64_t M, int N, int K,
                                                        int mchunk, float* __restrict__ part) {
    __shared__ float Ds[16][68];
    __shared__ float As[16][68];
    const int ntk = (K + 63) / 64;
    const int n0 = (blockIdx.x / ntk) * 64, k0 = (blockIdx.x % ntk) * 64;
    const int64_t mbeg = (int64_t)blockIdx.y * mchunk;
    const int64_t mend = mbeg + mchunk < M ? mbeg + mchunk : M;
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const bool do_bias = db != nullptr && k0 == 0;
    float bsum = 0.f;
    const int lm = tid >> 4, lc = (tid & 15) * 4;   // loader: row lm (0..15), 4 columns at lc
    float acc[4][4] = {};
    for (int64_t m0 = mbeg; m0 < mend; m0 += 16) {
        const int64_t m = m0 + lm;
        float4 dv = make_float4(0.f, 0.f, 0.f, 0.f), av = dv;
        if (m < mend) {
            const float sc = row_scale ? row_scale[m / rps] : 1.0f;
            if (n0 + lc + 3 < N) { dv = load4(dy + m * ldd + n0 + lc); }
            else { float t[4] = {0, 0, 0, 0}; for (int e = 0; e < 4; ++e) if (n0 + lc + e < N) t[e] = dy[m * ldd + n0 + lc + e]; dv = make_float4(t[0], t[1], t[2], t[3]); }
            dv.x *= sc; dv.y *= sc; dv.z *= sc; dv.w *= sc;
            if (k0 + lc + 3 < K) { av = load4(a + m * lda + k0 + lc); }
            else { float t[4] = {0, 0, 0, 0}; for (int e = 0; e < 4; ++e) if (k0 + lc + e < K) t[e] = a[m * lda + k0 + lc + e]; av = make_float4(t[0], t[1], t[2], t[3]); }
        }
        __syncthreads();
        *reinterpret_cast<float4*>(&Ds[lm][lc]) = dv;
        *reinterpret_cast<float4*>(&As[lm][lc]) = av;
        __syncthreads();
        if (do_bias && tid < 64) {
#pragma unroll
            for (int mm = 0; mm < 16; ++mm) bsum += Ds[mm][tid];
        }
#pragma unroll
        for (int mm = 0; mm < 16; ++mm) {
            const float4 d4 = *reinterpret_cast<const float4*>(&Ds[mm][ty * 4]);
            const float4 a4 = *reinterpret_cast<const float4*>(&As[mm][tx * 4]);
            const float dr[4] = {d4.x, d4.y, d4.z, d4.w};
            const float ar[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(dr[i], ar[j], acc[i][j]);
        }
    }
    {
        float* oW = part + (int64_t)blockIdx.y * ((int64_t)N * K + N);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + ty * 4 + i, k = k0 + tx * 4 + j;
                if (n < N && k < K) oW[(int64_t)n * K + k] = acc[i][j];
            }
        if (do_bias && tid < 64 && n0 + tid < N) oW[(int64_t)N * K + n0 + tid] = bsum;
    }
}

// dW[i] += sum_c part[c][i] (i < N*K), db[n] += sum_c part[c][N*K + n], chunks added in index order: the fixed-order second stage
// of the slab form of the three kernels above
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int nch, int64_t stride, float* __restrict__ dW,
                                                           float* __restrict__ db, int64_t nk, int n) {
    const int64_t tot4 = (nk + n) / 4;         // N, K multiples of 4 in every caller (checked by the launcher)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot4; i += (int64_t)gridDim.x * 256) {
        float4 s = *reinterpret_cast<const float4*>(part + 4 * i);
        for (int c = 1; c < nch; ++c) {
            const float4 t = *reinterpret_cast<const float4*>(part + c * stride + 4 * i);
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        float* o = 4 * i < nk ? dW + 4 * i : (db ? db + (4 * i - nk) : nullptr);
        if (o) {
            float4 cur = *reinterpret_cast<const float4*>(o);
            cur.x += s.x; cur.y += s.y; cur.z += s.z; cur.w += s.w;
            *reinterpret_cast<float4*>(o) = cur;
        }
    }
}

// First level of the two-level form (many slabs, few elements: e.g. the 96 -> 192 skip projection of block 1 leaves 784 slabs of
// 18,624 floats, which 19 workgroups of the kernel above take 280 us to walk): workgroup (x, g) adds the `group` consecutive slabs
// of group g in index order into the group's first slab, in place; wgrad_reduce_kernel then adds the group sums (slab step =
// group).  The grouping depends only on (nch, element count), so the summation tree is fixed: still bit-reproducible.
__global__ __launch_bounds__(256) void wgrad_reduce_group_kernel(float* __restrict__ part, int nch, int group, int64_t stride, int64_t tot4) {
    const int c0 = blockIdx.y * group, c1 = min(nch, c0 + group);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot4; i += (int64_t)gridDim.x * 256) {
        float4 s = *reinterpret_cast<const float4*>(part + c0 * stride + 4 * i);
        int c = c0 + 1;
        for (; c + 3 < c1; c += 4) {      // four loads in flight, added in index order
            const float4 t0 = *reinterpret_cast<const float4*>(part + c * stride + 4 * i);
            const float4 t1 = *reinterpret_cast<const float4*>(part + (c + 1) * stride + 4 * i);
            const float4 t2 = *reinterpret_cast<const float4*>(part + (c + 2) * stride + 4 * i);
            const float4 t3 = *reinterpret_cast<const float4*>(part + (c + 3) * stride + 4 * i);
            s.x += t0.x; s.y += t0.y; s.z += t0.z; s.w += t0.w;
            s.x += t1.x; s.y += t1.y; s.z += t1.z; s.w += t1.w;
            s.x += t2.x; s.y += t2.y; s.z += t2.z; s.w += t2.w;
            s.x += t3.x; s.y += t3.y; s.z += t3.z; s.w += t3.w;
        }
        for (; c < c1; ++c) {
            const float4 t = *reinterpret_cast<const float4*>(part + c * stride + 4 * i);
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        *reinterpret_cast<float4*>(part + c0 * stride + 4 * i) = s;
    }
}

// wgrad_pp.hip: the ping-pong weight gradient (256 n x 192 k tile per workgroup, 64-row steps over an M chunk)
bool mvit_internal_wgrad_pp_plan(int64_t lda, int64_t ldd, int64_t M, int N, int K, int64_t* nch, int* mchunk);
int mvit_internal_wgrad_pp(const void* a, int64_t lda, const void* dy, int64_t ldd, float* part, int64_t M, int N, int K, int64_t nch, int mchunk,
                           int do_bias, hipStream_t st);

// Launch plan shared by the entry point and the workspace query.
struct WgradPlan { int path; int64_t nch; int mchunk; };     // path: 0 fp32 VALU, 1 big MFMA tile, 2 96x96 MFMA tile, 3 ping-pong tile, < 0 error
static WgradPlan wgrad_plan(int a_dtype, int64_t lda, int dy_dtype, int64_t ldd, bool scaled, int64_t M, int N, int K, int act_dtype) {
    WgradPlan p = {MVIT_EUNSUPPORTED, 0, 0};
    // M chunk per workgroup: enough workgroups to fill the chip (~2048), at least 1024 rows
    const int64_t tiles = act_dtype == MVIT_F32 ? (int64_t)((N + 63) / 64) * ((K + 63) / 64) : (int64_t)(N / 96) * (K / 96);
    int64_t mc = (M * (tiles > 0 ? tiles : 1) + 2047) / 2048;
    mc = mc < WG_MCHUNK ? WG_MCHUNK : mc;
    mc = (mc + 63) / 64 * 64;
    const int64_t mchunks = (M + mc - 1) / mc;
    if (mchunks > 65535) { p.path = MVIT_EINVAL; return p; }
    if (act_dtype == MVIT_F32) {
        if (a_dtype != MVIT_F32 || dy_dtype != MVIT_F32) { p.path = MVIT_EDTYPE; return p; }
        if ((lda & 3) || (ldd & 3)) return p;
        p.path = 0; p.nch = mchunks; p.mchunk = (int)mc;
        return p;
    }
    if (act_dtype != MVIT_BF16) { p.path = MVIT_EDTYPE; return p; }
    if (N % 96 || K % 96 || (lda & 7) || (ldd & 7)) return p;
    if (a_dtype == MVIT_BF16 && dy_dtype == MVIT_BF16 && !scaled) {
        int64_t nch_pp = 0;
        int mch_pp = 0;
        if (mvit_internal_wgrad_pp_plan(lda, ldd, M, N, K, &nch_pp, &mch_pp)) { p.path = 3; p.nch = nch_pp; p.mchunk = mch_pp; return p; }
    }
    static const bool use_big = getenv("MVIT_WGRAD_NO_BIG") == nullptr;
    if (use_big && a_dtype == MVIT_BF16 && dy_dtype == MVIT_BF16 && !scaled && N % 32 == 0 && K % 32 == 0 && N >= 32 && K >= 32 && M % 64 == 0 &&
        64 * lda < (1ll << 31) && 64 * ldd < (1ll << 31)) {
        const int64_t bt = (int64_t)((N + WB_BP - 1) / WB_BP) * ((K + WB_BQ - 1) / WB_BQ);
        // M chunks: fewer, longer chunks (each ends with a 96 KiB fp32 tile going to its slab) as long as the grid stays inside ONE
        // round of resident workgroups (512 slots; 528 workgroups ran 17 % slower than 456 on the fc1 shape).  Alone on the chip the
        // best count is the largest that fits (profiles/r2_wgrad_wgs_sweep.txt: 456-504 workgroups, -5 %); inside the step, where
        // these GEMMs share the chip with the data-gradient stream, ~384 (256 for layers with <= 8 tiles) measured 0.1 ms better
        // than that (3 x 3 runs in one session), so it stays.  MVIT_WGRAD_WGS overrides.
        static const int wg_env = getenv("MVIT_WGRAD_WGS") ? atoi(getenv("MVIT_WGRAD_WGS")) : 0;
        const int wg_target = wg_env > 0 ? wg_env : (bt <= 8 ? 256 : 384);
        int64_t nch = (wg_target + bt - 1) / bt;
        int64_t bmc = ((M / 64 + nch - 1) / nch) * 64;          // rows per chunk, multiple of 64
        if (bmc < 512) bmc = 512;
        p.path = 1; p.nch = (M + bmc - 1) / bmc; p.mchunk = (int)bmc;
        return p;
    }
    p.path = 2; p.nch = mchunks; p.mchunk = (int)mc;
    return p;
}

// fp32 workspace bytes mvit_linear_wgrad needs for this problem; 0 if the shape is unsupported
extern "C" int64_t mvit_linear_wgrad_workspace_bytes(int a_dtype, int64_t lda, int dy_dtype, int64_t ldd, int has_row_scale, int64_t M,
                                                     int N, int K, int act_dtype) {
    const WgradPlan p = wgrad_plan(a_dtype, lda, dy_dtype, ldd, has_row_scale != 0, M, N, K, act_dtype);
    if (p.path < 0 || (N & 3) || (K & 3)) return 0;
    return p.nch * ((int64_t)N * K + N) * (int64_t)sizeof(float);
}

// dW (and db when given: db[n] += sum_m scale*dy[m][n]) must be zeroed (or hold the value to accumulate onto) by the caller.
// Every M chunk writes its partial dW / db to its own slab of `workspace` (>= mvit_linear_wgrad_workspace_bytes) and a second kernel
// adds the slabs in chunk order -- bit-reproducible; there is no float-atomic form.
extern "C" int mvit_linear_wgrad(const void* a, int a_dtype, int64_t lda, const void* dy, int dy_dtype, int64_t ldd,
                                 const float* row_scale, int64_t rows_per_scale, float* dW, float* db, int64_t M, int N,
                                 int K, int act_dtype, float* workspace, int64_t workspace_bytes, void* stream) {
    if (!a || !dy || !dW || M <= 0 || N <= 0 || K <= 0) return MVIT_EINVAL;
    if (row_scale && rows_per_scale <= 0) return MVIT_EINVAL;
    hipStream_t st = as_stream(stream);
    const WgradPlan p = wgrad_plan(a_dtype, lda, dy_dtype, ldd, row_scale != nullptr, M, N, K, act_dtype);
    if (p.path < 0) return p.path;
    const int64_t stride = (int64_t)N * K + N;
    if ((N & 3) || (K & 3)) return MVIT_EUNSUPPORTED;
    if (!workspace || workspace_bytes < p.nch * stride * (int64_t)sizeof(float)) return MVIT_EINVAL;
    float* part = workspace;
    const int mchunk = p.mchunk;
    if (p.path == 3) {
        const int rc = mvit_internal_wgrad_pp(a, lda, dy, ldd, part, M, N, K, p.nch, mchunk, db != nullptr, st);
        if (rc != MVIT_OK) return rc;
    } else if (p.path == 0) {
        dim3 grid(((N + 63) / 64) * ((K + 63) / 64), (unsigned)p.nch);
        hipLaunchKernelGGL(wgrad_f32_kernel, grid, dim3(256), 0, st, (const float*)a, lda, (const float*)dy, ldd, row_scale,
                           rows_per_scale, dW, db, M, N, K, mchunk, part);
    } else if (p.path == 1) {
        static DevFlags attr_done_tab; DevFlag attr_done = dev_flag(attr_done_tab);
        if (!attr_done) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_big_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, WB_SMEM) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_big_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, WB_SMEM) != hipSuccess)
                return MVIT_ELAUNCH;
            attr_done = true;
        }
        const int64_t bt = (int64_t)((N + WB_BP - 1) / WB_BP) * ((K + WB_BQ - 1) / WB_BQ);
        dim3 bgrid((unsigned)bt, (unsigned)p.nch);
        static const bool khalf_ok = getenv("MVIT_WGRAD_NO_KHALF") == nullptr;
        if (K <= 96 && khalf_ok)
            hipLaunchKernelGGL(wgrad_big_kernel<true>, bgrid, dim3(256), WB_SMEM, st, (const bf16_t*)a, lda, (const bf16_t*)dy, ldd, dW, db, M, N, K,
                               mchunk, part);
        else
            hipLaunchKernelGGL(wgrad_big_kernel<false>, bgrid, dim3(256), WB_SMEM, st, (const bf16_t*)a, lda, (const bf16_t*)dy, ldd, dW, db, M, N, K,
                               mchunk, part);
    } else {
        dim3 grid((N / 96) * (K / 96), (unsigned)p.nch);
#define WG(TA, TD) \
        hipLaunchKernelGGL((wgrad_mfma_kernel<TA, TD>), grid, dim3(192), 0, st, (const TA*)a, lda, (const TD*)dy, ldd, row_scale, rows_per_scale, dW, db, M, N, K, mchunk, part)
        if (a_dtype == MVIT_BF16 && dy_dtype == MVIT_BF16) WG(bf16_t, bf16_t);
        else if (a_dtype == MVIT_BF16 && dy_dtype == MVIT_F32) WG(bf16_t, float);
        else if (a_dtype == MVIT_F32 && dy_dtype == MVIT_F32) WG(float, float);
        else if (a_dtype == MVIT_F32 && dy_dtype == MVIT_BF16) WG(float, bf16_t);
        else return MVIT_EDTYPE;
#undef WG
    }
    MVIT_LAUNCH_CHECK();
    {
        int64_t blocks = ((stride / 4) + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        int nslab = (int)p.nch;
        int64_t step = stride;
        if (blocks < 256 && nslab >= 16) {          // few elements, many slabs: spread the slabs over the chip first
            int groups = (int)((512 + blocks - 1) / blocks);
            if (groups > nslab / 4) groups = nslab / 4;
            const int group = (nslab + groups - 1) / groups;
            groups = (nslab + group - 1) / group;
            hipLaunchKernelGGL(wgrad_reduce_group_kernel, dim3((unsigned)blocks, (unsigned)groups), dim3(256), 0, st, part, nslab, group, stride,
                               stride / 4);
            MVIT_LAUNCH_CHECK();
            nslab = groups;
            step = stride * group;
        }
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, part, nslab, step, dW, db, (int64_t)N * K, N);
        MVIT_LAUNCH_CHECK();
    }
    return MVIT_OK;
}

