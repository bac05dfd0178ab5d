// Ping-pong weight gradient: part[chunk][n][k] = sum over the chunk's rows m of dy[m][n] * a[m][k]  (16-bit operands, fp32 slabs),
// the long-contraction product of the training step (M = 12,544 ... 200,704 rows against N x K <= 3072 x 768 outputs).
//   reference: torch.autograd's mm backward w.r.t. the weight behind every nn.Linear of the block
//   (slowfast/models/attention.py:231,281, slowfast/models/common.py:27-31; tools/train_net.py:231).
//
// The skeleton is linear_pp.hip's (one 512-thread workgroup per CU, two wave groups staggered by one barrier, eight segments
// per 64-row step of the contraction, LDS-DMA ring closed by counted vmcnt -- the schedule and its hazard argument are written
// out there and hold here unchanged: the units have the same sizes and the same readers).  What differs:
//   * the contraction index m is the ROW index of both operands, so a step's units are row-major [64 m][columns] images:
//     X0 / X1 = dy[:, n0 .. n0+127] / [n0+128 .. n0+255] (256-byte rows, in the place of T0 / T1), Ya = a[:, k0 .. k0+127]
//     (256-byte rows) and Yb = a[:, k0+128 .. k0+191] (128-byte rows) in the place of the weight tile;
//   * MFMA fragments want 8 consecutive m of ONE column, so they are read with ds_read_b64_tr_b16 (two per fragment: rows
//     8 lg .. +3 and +4 .. +7 of the 32-row k-step); 16-byte chunk ch of row r sits at ch ^ (((r & 3) << 2) | ((r >> 2) & 3)) in the
//     256-byte-row images (cdna guide T10, image (b)) and at ch ^ ((3 * ((r >> 1) & 7)) & 7) in the 128-byte-row image: both
//     conflict-free for this read pattern (tools/probes/bank_tr.py enumerates the banks);
//   * A operand = the a columns (k), B operand = the dy columns (n): a lane ends with 4 consecutive k of one n, a float4 of
//     the slab row part[n][k..k+3]; the bias gradient (column sums of dy) rides on 4 extra MFMAs per step against a ones fragment;
//   * one (256 n x 192 k tile, M chunk) per workgroup, no persistence: tiles x chunks <= 256.
// Chunk slabs are added in chunk order by wgrad_reduce_kernel (linear_bwd.hip): bit-reproducible like every gradient of the step.
#include <stdlib.h>

#include "common.h"

#define WP_T_HALF 16384                  // 64 rows x 256 B
#define WP_YA_OFF 32768
#define WP_YB_OFF (WP_YA_OFF + 16384)    // 64 rows x 128 B
#define WP_BUF 57344
#define WP_SMEM (2 * WP_BUF)

__device__ __forceinline__ f32x4 wp_mfma(bf16x8 a, bf16x8 b, f32x4 c) {
#ifdef MVIT_HALF_IS_FP16
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mfma16_t, a), __builtin_bit_cast(mfma16_t, b), c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma16_t, a), __builtin_bit_cast(mfma16_t, b), c, 0, 0, 0);
#endif
}
__device__ __forceinline__ bf16x4 wp_tr(uint32_t addr) {
    bf16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
__device__ __forceinline__ void wp_dma(const char* base, uint32_t off, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(off), "s"(base) : "memory");
}
__device__ __forceinline__ bf16x8 wp_join(bf16x4 lo, bf16x4 hi) {
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}
#define WP_BARRIER() asm volatile("s_barrier" ::: "memory")
#define WP_SB() __builtin_amdgcn_sched_barrier(0)

__global__ __launch_bounds__(512, 2) void wgrad_pp_kernel(const bf16_t* __restrict__ a, int64_t lda, const bf16_t* __restrict__ dy, int64_t ldd,
                                                          float* __restrict__ part, int64_t M, int N, int K, int mchunk, int do_bias) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int grp = wave >> 2, wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, lg = lane >> 4;
    const int q4 = l15 >> 2, p4 = l15 & 3;

    const int ntk = K / 192;
    const int tile = blockIdx.x, chunk = blockIdx.y;
    const int n0 = (tile / ntk) * 256, k0 = (tile % ntk) * 192;
    const int64_t mbeg = (int64_t)chunk * mchunk, mend = mbeg + mchunk < M ? mbeg + mchunk : M;
    const int nk = (int)((mend - mbeg) / 64);
    if (nk <= 0) return;

    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    // ---- fragment addresses in buffer 0, k-step 0 (k-step 1: + 32 rows) -------------------------------------------------------------
    // transposing read h2 of a fragment: lane (lg, 4 q + p) addresses row 8 lg + 4 h2 + q, columns c0 + 4 p .. + 3 of its block
    uint32_t xa[4][2], ya[6][2];           // [column block of 16][h2]
    uint32_t yks[6];                       // byte step of one k-step (32 rows) in the image of that block: 8192 or 4096
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
        const int row = 8 * lg + 4 * h2 + q4;
        const int sA = ((row & 3) << 2) | ((row >> 2) & 3), sB = (3 * ((row >> 1) & 7)) & 7;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {      // dy columns 64 (wm & 1) + 16 mt .. of the wave's X half
            const int ch = 8 * (wm & 1) + 2 * mt + (p4 >> 1);
            xa[mt][h2] = lds0 + grp * WP_T_HALF + 256 * row + 16 * (ch ^ sA) + 8 * (p4 & 1);
        }
#pragma unroll
        for (int nt = 0; nt < 6; ++nt) {      // a columns 96 wn + 16 nt ..: the first 128 of the tile live in Ya, the last 64 in Yb
            const int col = 96 * wn + 16 * nt;
            if (col < 128) {
                const int ch = col / 8 + (p4 >> 1);
                ya[nt][h2] = lds0 + WP_YA_OFF + 256 * row + 16 * (ch ^ sA) + 8 * (p4 & 1);
            } else {
                const int ch = (col - 128) / 8 + (p4 >> 1);
                ya[nt][h2] = lds0 + WP_YB_OFF + 128 * row + 16 * (ch ^ sB) + 8 * (p4 & 1);
            }
        }
    }
#pragma unroll
    for (int nt = 0; nt < 6; ++nt) yks[nt] = __builtin_amdgcn_readfirstlane((96 * wn + 16 * nt) < 128 ? 8192 : 4096);

    // ---- DMA pieces (1 KiB): X halves and Ya: 4 rows x 256 B (lane -> row lane / 16, position lane % 16), pieces 2 wave + i;
    //      Yb: 8 rows x 128 B (lane -> row lane / 8, position lane % 8), piece = wave.  Source chunk = position ^ swizzle(row). ----
    uint32_t xo[2], yo[2], yob;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 4 * (2 * wave + i) + (lane >> 4);
        const int c = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
        xo[i] = (uint32_t)(row * (int)ldd + 8 * c) * 2u;
        yo[i] = (uint32_t)(row * (int)lda + 8 * c) * 2u;
    }
    {
        const int row = 8 * wave + (lane >> 3);
        const int c = (lane & 7) ^ ((3 * ((row >> 1) & 7)) & 7);
        yob = (uint32_t)(row * (int)lda + 128 + 8 * c) * 2u;
    }
    // the second dy half of a tile that hangs over N repeats the first (its accumulators are never stored)
    const int x1_cols = n0 + 128 < N ? 128 : 0;
    const char* xp = reinterpret_cast<const char*>(dy + mbeg * ldd + n0);       // K-tile 0; + 64 rows per step
    const char* yp = reinterpret_cast<const char*>(a + mbeg * lda + k0);
    const int64_t xstep = 64 * ldd * 2, ystep = 64 * lda * 2;
    const uint32_t d_t = lds0 + 1024 * (2 * wave), d_ya = lds0 + WP_YA_OFF + 1024 * (2 * wave), d_yb = lds0 + WP_YB_OFF + 1024 * wave;
    // cursors: pointers of K-tile G+1 (c1) and G+2 (c2); past the last K-tile they stay (harmless re-reads into consumed buffers)
    int kt1 = nk > 1 ? 1 : 0, kt2 = nk > 2 ? 2 : kt1;
    auto xptr = [&](int kt, int half) { return xp + kt * xstep + half * (x1_cols * 2); };
    auto yptr = [&](int kt) { return yp + kt * ystep; };
    auto dma_t = [&](int kt, int half, uint32_t bo) {
        const char* b_ = xptr(kt, half);
        wp_dma(b_, xo[0], d_t + half * WP_T_HALF + bo);
        wp_dma(b_, xo[1], d_t + half * WP_T_HALF + bo + 1024);
    };
    auto dma_wa = [&](int kt, uint32_t bo) {
        const char* b_ = yptr(kt);
        wp_dma(b_, yo[0], d_ya + bo);
        wp_dma(b_, yo[1], d_ya + bo + 1024);
    };
    auto dma_wb = [&](int kt, uint32_t bo) { wp_dma(yptr(kt), yob, d_yb + bo); };

    dma_wa(0, 0); dma_t(0, 0, 0); dma_t(0, 1, 0); dma_wb(0, 0);
    dma_wa(kt1, WP_BUF); dma_t(kt1, 0, WP_BUF);

    // acc[mt][nt]: n = n0 + 64 wm + 16 mt + l15 ; k = k0 + 96 wn + 16 nt + 4 lg + (0..3).  bacc[mt]: column sums of dy (all 16 rows equal)
    f32x4 acc[4][6], bacc[2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 6; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    bacc[0] = bacc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = MVIT_ONE16;

    bf16x4 tlo[2][2][2], thi[2][2][2], wlo[2][3][2], whi[2][3][2];      // [set][block][k-step], low / high 4 rows of a fragment
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // K-tile 0 has landed (Wa(1), T0(1) may still fly)
    WP_BARRIER();
#define WP_RD_T(SET, KS, BO) _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_) { \
        tlo[SET][m_][KS] = wp_tr(xa[2 * SET + m_][0] + (BO) + 8192 * KS); thi[SET][m_][KS] = wp_tr(xa[2 * SET + m_][1] + (BO) + 8192 * KS); }
#define WP_RD_W(SET, KS, BO) _Pragma("unroll") for (int n_ = 0; n_ < 3; ++n_) { \
        wlo[SET][n_][KS] = wp_tr(ya[3 * SET + n_][0] + (BO) + yks[3 * SET + n_] * KS); whi[SET][n_][KS] = wp_tr(ya[3 * SET + n_][1] + (BO) + yks[3 * SET + n_] * KS); }
    WP_RD_T(0, 0, 0u) WP_RD_T(0, 1, 0u)
    if (grp) WP_BARRIER();                                // the stagger: group 1 runs one slot behind group 0
    WP_SB();

#define WP_MM(MSET, NSET, TS, WS) \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int nt_ = 0; nt_ < 3; ++nt_) _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) \
        acc[2 * MSET + mt][3 * NSET + nt_] = wp_mfma(wp_join(wlo[WS][nt_][ks], whi[WS][nt_][ks]), wp_join(tlo[TS][mt][ks], thi[TS][mt][ks]), acc[2 * MSET + mt][3 * NSET + nt_]);
#define WP_WAIT_W(S) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wlo[S][0][0]), "+v"(wlo[S][0][1]), "+v"(wlo[S][1][0]), "+v"(wlo[S][1][1]), "+v"(wlo[S][2][0]), "+v"(wlo[S][2][1])); \
        asm volatile("" : "+v"(whi[S][0][0]), "+v"(whi[S][0][1]), "+v"(whi[S][1][0]), "+v"(whi[S][1][1]), "+v"(whi[S][2][0]), "+v"(whi[S][2][1])); }
#define WP_WAIT_T(S) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tlo[S][0][0]), "+v"(tlo[S][0][1]), "+v"(tlo[S][1][0]), "+v"(tlo[S][1][1])); \
        asm volatile("" : "+v"(thi[S][0][0]), "+v"(thi[S][0][1]), "+v"(thi[S][1][0]), "+v"(thi[S][1][1])); }
#define WP_MSEG(BODY) { asm volatile("s_setprio 1"); WP_SB(); BODY WP_SB(); asm volatile("s_setprio 0"); WP_BARRIER(); WP_SB(); }

    for (int g = 0; g < nk; ++g) {
        const uint32_t bo = (g & 1) ? WP_BUF : 0, nbo = WP_BUF - bo;
        // ---- L0: w0(G) ; DMA T1(G+1) ----------------------------------------------------------------------------
        WP_RD_W(0, 0, bo) WP_RD_W(0, 1, bo)
        dma_t(kt1, 1, nbo);
        WP_SB(); WP_BARRIER();
        WP_WAIT_W(0)
        WP_SB();
        WP_MSEG(WP_MM(0, 0, 0, 0))
        // ---- L1: w1(G) ; DMA Wb(G+1) ; reads retired BEFORE the barrier (Wa(G+2) is issued in the next slot) -------
        WP_RD_W(1, 0, bo) WP_RD_W(1, 1, bo)
        dma_wb(kt1, nbo);
        WP_WAIT_W(1)
        WP_SB(); WP_BARRIER(); WP_SB();
        WP_MSEG(WP_MM(0, 1, 0, 1))
        // ---- L2: t1(G) ; DMA Wa(G+2) ; T0(G+1), T1(G+1) landed --------------------------------------------------------
        WP_RD_T(1, 0, bo) WP_RD_T(1, 1, bo)
        dma_wa(kt2, bo);
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        WP_SB(); WP_BARRIER();
        WP_WAIT_T(1)
        WP_SB();
        // (bias gradient: the wave pair of a dy column range shares it -- wn = 0 sums blocks 0, 1 = set 0, wn = 1 blocks 2, 3 = set 1;
        //  set 1 is complete here, set 0 was read one segment pair earlier)
        WP_MSEG(WP_MM(1, 1, 1, 1)
                if (do_bias) { _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) {
                    const bf16x8 tb = wn ? wp_join(tlo[1][mt][ks], thi[1][mt][ks]) : wp_join(tlo[0][mt][ks], thi[0][mt][ks]);
                    bacc[mt] = wp_mfma(ones, tb, bacc[mt]); } })
        // ---- L3: t0(G+1) ; DMA T0(G+2) ; W(G+1) landed -------------------------------------------------------------------
        WP_RD_T(0, 0, nbo) WP_RD_T(0, 1, nbo)
        dma_t(kt2, 0, bo);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        WP_SB(); WP_BARRIER();
        WP_WAIT_T(0)
        WP_SB();
        WP_MSEG(WP_MM(1, 0, 1, 0))
        kt1 = kt2;
        kt2 = kt2 + 1 < nk ? kt2 + 1 : kt2;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA may outlive the workgroup
    if (!grp) WP_BARRIER();                               // group 0 is one barrier short of group 1

    // ---- this chunk's tile to its own slab: part[chunk][n][k] (+ [N*K + n] for the column sums of dy) ---------------------------
    float* oW = part + (int64_t)chunk * ((int64_t)N * K + N);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int n = n0 + 64 * wm + 16 * mt + l15;
        if (n < N) {
#pragma unroll
            for (int nt = 0; nt < 6; ++nt)
                *reinterpret_cast<f32x4*>(oW + (int64_t)n * K + k0 + 96 * wn + 16 * nt + 4 * lg) = acc[mt][nt];
        }
    }
    if (do_bias && k0 == 0 && lg == 0) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int n = n0 + 64 * wm + 16 * (2 * wn + mt) + l15;
            if (n < N) oW[(int64_t)N * K + n] = bacc[mt][0];
        }
    }
#undef WP_RD_T
#undef WP_RD_W
#undef WP_MM
#undef WP_WAIT_W
#undef WP_WAIT_T
#undef WP_MSEG
}

// Shapes the kernel takes and its launch plan: one workgroup per (tile, chunk), at most 256 workgroups
bool mvit_internal_wgrad_pp_plan(int64_t lda, int64_t ldd, int64_t M, int N, int K, int64_t* nch, int* mchunk) {
    static const char* env = getenv("MVIT_WGRAD_PP");
    if (env && env[0] == '0') return false;
    if (K % 192 || N % 128 || (lda & 7) || (ldd & 7) || M % 64 || 64 * lda >= (1ll << 30) || 64 * ldd >= (1ll << 30)) return false;
    // default routing (profiles/r3_wgrad_ab.txt): ahead of the 128 x 192 kernel where no tile hangs over N (fc1 1536 x 384: 85 vs 96 us,
    // stage 4, the 768 x 192 layers of stage 2: 109 vs 156 us), level or behind where a quarter of the tile is idle (N = 384, 1152)
    if (!(env && env[0] == '1') && (M < 8192 || N % 256)) return false;
    const int64_t tiles = (int64_t)((N + 255) / 256) * (K / 192);
    if (tiles > 256) return false;
    int64_t c = 256 / tiles;
    int64_t rows = ((M / 64 + c - 1) / c) * 64;
    if (rows < 512) rows = 512;
    *mchunk = (int)rows;
    *nch = (M + rows - 1) / rows;
    return true;
}

int mvit_internal_wgrad_pp(const void* a, int64_t lda, const void* dy, int64_t ldd, float* part, int64_t M, int N, int K, int64_t nch, int mchunk,
                           int do_bias, hipStream_t st) {
    static DevFlags attr_done_tab; DevFlag attr_done = dev_flag(attr_done_tab);
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_pp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, WP_SMEM) != hipSuccess)
            return MVIT_ELAUNCH;
        attr_done = true;
    }
    dim3 grid((unsigned)(((N + 255) / 256) * (K / 192)), (unsigned)nch);
    hipLaunchKernelGGL(wgrad_pp_kernel, grid, dim3(512), WP_SMEM, st, (const bf16_t*)a, lda, (const bf16_t*)dy, ldd, part, M, N, K, mchunk, do_bias);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}
