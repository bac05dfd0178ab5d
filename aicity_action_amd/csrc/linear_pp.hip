// Ping-pong GEMM for the long-row linear layers: y = epilogue(a . w^T), 16-bit operands, fp32 accumulate.
//
// One persistent 512-thread workgroup per CU walks 256 (tokens) x 192 (weight rows) output tiles; (tile, K-tile) is ONE stream
// of 64-wide K-tiles, so the LDS-DMA ring runs across tile boundaries and a tile's epilogue overlaps the next tile's loads.
//
//   * 8 waves = 4 (tokens) x 2 (weight rows), each owning 64 x 96 of the tile as 4 x 6 accumulators of
//     v_mfma_f32_16x16x32 (the product is computed transposed -- A operand = weight rows, B operand = token rows -- so a lane
//     owns ONE token and 4 consecutive output columns per accumulator: bias / GELU / residual run in registers and rows
//     leave in 16-byte pieces; 16-bit outputs pair two accumulators with v_permlane16_swap).
//   * LDS: two K-tile buffers of {T0, T1: 128 token rows x 128 B each | W: 192 weight rows x 128 B} = 56 KiB each, filled
//     by global_load_lds_dwordx4 (1 KiB pieces = 8 rows x 128 B; the bank swizzle chunk ^= (row >> 1) & 7 is applied to the
//     per-lane SOURCE address, the image itself is lane-linear), 7 pieces per wave and K-tile.
//   * The two wave groups (waves 0-3: token rows 0-127, waves 4-7: rows 128-255; SIMD partners are always in different
//     groups) run the same stream of segments, staggered by one barrier: a K-tile is four load segments L0..L3 (4-6
//     fragment reads + 1-2 DMA pieces) alternating with four matrix segments M0..M3 (12 MFMAs = one 32 x 48 quadrant of the
//     wave tile x K = 64), so while one group multiplies, its SIMD partners read fragments and issue DMA.
//
// Hazards are closed by count, not by timing (cdna guide section 5, "Read a staged buffer one phase AFTER the wait that retires
// it"): with G the global K-tile index and b = G & 1 its buffer, a wave issues
//     L(G,0): reads w0(G)            DMA T1(G+1) -> b^1        L(G,2): reads t1(G)     DMA Wa(G+2) -> b, then vmcnt(3)
//     L(G,1): reads w1(G)            DMA Wb(G+1) -> b^1        L(G,3): reads t0(G+1)   DMA T0(G+2) -> b, then vmcnt(4)
// (w0 / w1: weight rows 0-47 / 48-95 of the wave, t0 / t1: token rows 0-31 / 32-63; Wa / Wb: weight rows 0-127 / 128-191).
// vmcnt(3) after L(G,2)'s issue leaves {Wb(G+1), Wa(G+2) x2} outstanding, i.e. T0(G+1) and T1(G+1) have landed for every
// wave before the barrier that precedes their first read (L(G,3) of either group); vmcnt(4) after L(G,3)'s issue leaves
// {Wa(G+2) x2, T0(G+2) x2}: W(G+1) has landed before L(G+1,0).  A unit is re-filled at the earliest two barriers after the
// lgkmcnt(0) that retired its last read (L(G,1) waits BEFORE its barrier, because Wa(G+2) is issued by the other group in
// the very next slot).  Epilogue loads / stores sit between two K-tiles and only ever make these waits stricter.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

#ifndef PP_ABL
#define PP_ABL 0     // timing ablations (tools/build_pp_abl.sh; results invalid): 1 no MFMA, 2 no in-loop DMA, 4 no fragment reads, 8 cycle stamps, 16 no s_setprio
#endif
#define PP_BM 256
#define PP_BN 192
#define PP_BK 64
#define PP_T_HALF 16384                  // 128 rows x 128 B
#define PP_W_OFF 32768
#define PP_WB_OFF (PP_W_OFF + 16384)
#define PP_BUF 57344                     // T0 | T1 | W
#define PP_TILES_BYTES (2 * PP_BUF)      // 114688, followed by the bias panel (N floats)

// epilogues
enum {
    PP_B16 = 0,        // 16-bit y = acc (+bias)
    PP_GELU16 = 1,     // 16-bit y = GELU(acc)
    PP_GELU_PRE = 2,   // 16-bit y = GELU(acc), y2 = acc
    PP_GELU_DER = 3,   // 16-bit y = GELU(acc), y2 = GELU'(acc)
    PP_F32 = 4,        // fp32  y = acc
    PP_F32_RES = 5,    // fp32  y = aux(fp32 residual) + acc
    PP_F32_RES_SC = 6, // fp32  y = aux + scale[row] * acc
    PP_DG_PRE = 7,     // 16-bit y = scale[row] * acc * GELU'(aux 16-bit pre-activation)   (scale optional)
    PP_DG_DER = 8,     // 16-bit y = scale[row] * acc * aux (16-bit saved derivative)
    PP_NEPI = 9
};

__device__ __forceinline__ f32x4 pp_mfma(bf16x8 a, bf16x8 b, f32x4 c) {
#ifdef MVIT_HALF_IS_FP16
    if (PP_ABL & 1) { asm volatile("; no mfma" : "+v"(c) : "v"(a), "v"(b)); return c; }
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mfma16_t, a), __builtin_bit_cast(mfma16_t, b), c, 0, 0, 0);
#else
    if (PP_ABL & 1) { asm volatile("; no mfma" : "+v"(c) : "v"(a), "v"(b)); return c; }
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma16_t, a), __builtin_bit_cast(mfma16_t, b), c, 0, 0, 0);
#endif
}

template <int OFF>
__device__ __forceinline__ bf16x8 pp_rd(uint32_t addr) {
    bf16x8 v;
    if (PP_ABL & 4) { asm volatile("; no read" : "=v"(v)); return v; }
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// one 1-KiB LDS-DMA piece: SGPR base + 32-bit lane offset, LDS destination in M0 (M0 cannot be named as a clobber; nothing else
// in this kernel depends on it: gfx9+ LDS instructions do not read M0)
__device__ __forceinline__ void pp_dma(const char* base, uint32_t off, uint32_t lds) {
    if (PP_ABL & 2) { asm volatile("; no dma %0 %1 %2" ::"s"(lds), "v"(off), "s"(base) : "memory"); return; }
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(off), "s"(base) : "memory");
}
#define PP_BARRIER() asm volatile("s_barrier" ::: "memory")
#define PP_SB() __builtin_amdgcn_sched_barrier(0)

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
#if PP_ABL & 8
__device__ float g_pp_stamps[256 * 4];
/* ablation builds only (tools/opbench.py); not part of the C-ABI */
extern "C" __attribute__((visibility("default"))) int mvit_debug_pp_stamps(float* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pp_stamps), sizeof(float) * 256 * 4) == hipSuccess ? 0 : -3; }
#endif

struct PPCur {          // DMA cursor: source of one K-tile
    const char* ap;     // a + m0 * lda + k   (wave-uniform)
    const char* wp;     // w + n0 * K + k
    int t, kt;
    int rmax;           // last existing row of the tile's 256 (255 unless the tile hangs over M)
};

template <int EPI, typename TO>
__global__ __launch_bounds__(512, 2) void linear_pp_kernel(
    const bf16_t* __restrict__ a, int64_t lda, const bf16_t* __restrict__ w, const float* __restrict__ bias,
    const void* __restrict__ aux, int64_t ldaux, const float* __restrict__ row_scale, int64_t rps,
    TO* __restrict__ y, TO* __restrict__ y2, int64_t ldy, int64_t M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int grp = wave >> 2, wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, lg = lane >> 4;

    const int ntn = N / PP_BN;
    const int nt = (int)((M + PP_BM - 1) / PP_BM) * ntn;
    const int per = gridDim.x >> 3, q = (nt + 7) >> 3, xcd = blockIdx.x & 7;
    const int t_end = min(nt, (xcd + 1) * q);
    const int t_first = xcd * q + (blockIdx.x >> 3);
    if (t_first >= t_end) return;
    const int nk = K / PP_BK;
    const int my_tiles = (t_end - t_first + per - 1) / per;

    float* sbias = reinterpret_cast<float*>(smem + PP_TILES_BYTES);
    for (int i = tid; i < N; i += 512) sbias[i] = bias ? bias[i] : 0.f;
    __syncthreads();        // no DMA in flight yet: a plain barrier is fine here

    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    // fragment read addresses in buffer 0 (k-steps 0 / 1): row l15 of a 16-row block, 16-B chunk (4 ks + lg) ^ swz(row)
    uint32_t ta[2], wa[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int fo = ((4 * ks + lg) ^ ((l15 >> 1) & 7)) * 16;
        ta[ks] = lds0 + grp * PP_T_HALF + (64 * (wm & 1) + l15) * 128 + fo;
        wa[ks] = lds0 + PP_W_OFF + (96 * wn + l15) * 128 + fo;
    }
    // DMA pieces of this wave (1 KiB = 8 rows x 128 B, lane -> row lane / 8, position lane % 8): T0 / T1 / Wa rows 16 wave + 8 i,
    // Wb row 128 + 8 wave.  Per-lane byte offsets; the token offsets assume a full tile (the ragged last row panel recomputes them)
    const int prow = 16 * wave + (lane >> 3);
    uint32_t wo[2], wob, to[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = prow + 8 * i, ch = (lane & 7) ^ ((row >> 1) & 7);
        wo[i] = (uint32_t)(row * K + 8 * ch) * 2u;
        to[i] = (uint32_t)(row * (int)lda + 8 * ch) * 2u;
    }
    {
        const int rowb = 128 + 8 * wave + (lane >> 3);
        wob = (uint32_t)(rowb * K + 8 * ((lane & 7) ^ ((rowb >> 1) & 7))) * 2u;
    }
    const uint32_t d_t = lds0 + 1024 * (2 * wave);                  // + half * PP_T_HALF + 1024 i + buffer
    const uint32_t d_wa = lds0 + PP_W_OFF + 1024 * (2 * wave);      // + 1024 i + buffer
    const uint32_t d_wb = lds0 + PP_WB_OFF + 1024 * wave;
    const int64_t half_bytes = 128 * lda * 2;                       // T1 rows = T0 rows + 128

    auto setup = [&](PPCur& c, int tile) {
        const int64_t m0 = (int64_t)(tile / ntn) * PP_BM;
        c.ap = reinterpret_cast<const char*>(a + m0 * lda);
        c.wp = reinterpret_cast<const char*>(w + (int64_t)(tile % ntn) * PP_BN * K);
        c.t = tile;
        c.kt = 0;
        const int64_t left = M - 1 - m0;
        c.rmax = left > 255 ? 255 : (int)left;
    };
    auto advance = [&](PPCur& c) {
        if (c.kt + 1 < nk) { ++c.kt; c.ap += 2 * PP_BK; c.wp += 2 * PP_BK; }
        else if (c.t + per < t_end) setup(c, c.t + per);
        // else: past the last K-tile of this workgroup -- the cursor stays (harmless re-reads into consumed buffers)
    };
    auto dma_t = [&](const PPCur& c, int half, uint32_t bo) {
        if (c.rmax >= 255) {
            const char* base = c.ap + half * half_bytes;
            pp_dma(base, to[0], d_t + half * PP_T_HALF + bo);
            pp_dma(base, to[1], d_t + half * PP_T_HALF + bo + 1024);
        } else {        // the ragged last row panel: rows past M re-read row M - 1 (never stored)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = 128 * half + prow + 8 * i, rr = row < c.rmax ? row : c.rmax;
                pp_dma(c.ap, (uint32_t)(rr * (int)lda + 8 * ((lane & 7) ^ ((row >> 1) & 7))) * 2u, d_t + half * PP_T_HALF + bo + 1024 * i);
            }
        }
    };
    auto dma_wa = [&](const PPCur& c, uint32_t bo) {
        pp_dma(c.wp, wo[0], d_wa + bo);
        pp_dma(c.wp, wo[1], d_wa + bo + 1024);
    };
    auto dma_wb = [&](const PPCur& c, uint32_t bo) { pp_dma(c.wp, wob, d_wb + bo); };

    PPCur c1, c2;
    {
        PPCur c0;
        setup(c0, t_first);
        c1 = c0; advance(c1);
        c2 = c1; advance(c2);
        dma_wa(c0, 0); dma_t(c0, 0, 0); dma_t(c0, 1, 0); dma_wb(c0, 0);
        dma_wa(c1, PP_BUF); dma_t(c1, 0, PP_BUF);
    }

    // acc[mt][nt]: token 64 wm + 16 mt + l15 ; columns 96 wn + 16 nt + 4 lg + (0..3)
    f32x4 acc[4][6];
    int t_cur = t_first;
    auto init_acc = [&](int tile) {
        const int n0 = (tile % ntn) * PP_BN;
#pragma unroll
        for (int nt_ = 0; nt_ < 6; ++nt_) {
            const float4 b = *reinterpret_cast<const float4*>(sbias + n0 + 96 * wn + 16 * nt_ + 4 * lg);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) { acc[mt][nt_][0] = b.x; acc[mt][nt_][1] = b.y; acc[mt][nt_][2] = b.z; acc[mt][nt_][3] = b.w; }
        }
    };
    init_acc(t_cur);

    // per-lane byte offsets of a wave's output rows: a wave-uniform base (SALU) + ONE constant lane offset
    const uint32_t lane_o16 = (uint32_t)(l15 * (int)ldy + 16 * (lg & 1) + 8 * (lg >> 1)) * 2u;     // 16-bit pieces: block nt + (lg & 1), columns 8 (lg >> 1)..
    auto st16 = [&](const char* base, int mt, int np, int lim, u32x4 v) {     // base / mt / np / lim wave-uniform
        char* p_ = const_cast<char*>(base) + ((int64_t)(16 * mt) * ldy + 32 * np) * 2;
        if (l15 + 16 * mt < lim) asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(lane_o16), "v"(v), "s"(p_) : "memory");   // the nop: hipcc may overwrite the data registers right behind an asm store
    };

    bf16x8 tf[2][2][2], wf[2][3][2];      // [set][block][k-step]
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // K-tile 0 has landed (Wa(1), T0(1) may still fly)
    PP_BARRIER();
    tf[0][0][0] = pp_rd<0>(ta[0]); tf[0][0][1] = pp_rd<0>(ta[1]);
    tf[0][1][0] = pp_rd<2048>(ta[0]); tf[0][1][1] = pp_rd<2048>(ta[1]);
    if (grp) PP_BARRIER();                                // the stagger: group 1 runs one slot behind group 0
    PP_SB();
#if PP_ABL & 8
    const uint64_t loop_c0 = __builtin_readcyclecounter(), loop_r0 = __builtin_readsteadycounter();
    uint64_t s_epi = 0;
#endif

#define PP_PRIO(x) if (!(PP_ABL & 16)) asm volatile("s_setprio " #x)
#define PP_MM(MSET, NSET, TS, WS) \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int nt_ = 0; nt_ < 3; ++nt_) _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) \
        acc[2 * MSET + mt][3 * NSET + nt_] = pp_mfma(wf[WS][nt_][ks], tf[TS][mt][ks], acc[2 * MSET + mt][3 * NSET + nt_]);
#define PP_WAIT_W(S) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[S][0][0]), "+v"(wf[S][0][1]), "+v"(wf[S][1][0]), "+v"(wf[S][1][1]), "+v"(wf[S][2][0]), "+v"(wf[S][2][1]))
#define PP_WAIT_T(S) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tf[S][0][0]), "+v"(tf[S][0][1]), "+v"(tf[S][1][0]), "+v"(tf[S][1][1]))
#define PP_MSEG(BODY) { PP_PRIO(1); PP_SB(); BODY PP_SB(); PP_PRIO(0); PP_BARRIER(); PP_SB(); }

    int g = 0;
    for (int ti = 0; ti < my_tiles; ++ti) {
      for (int kt_cur = 0; kt_cur < nk; ++kt_cur, ++g) {
        const uint32_t bo = (g & 1) ? PP_BUF : 0, nbo = PP_BUF - bo;
        // ---- L0: w0(G) ; DMA T1(G+1) ----------------------------------------------------------------------------
        {
            const uint32_t w0 = wa[0] + bo, w1 = wa[1] + bo;
            wf[0][0][0] = pp_rd<0>(w0); wf[0][0][1] = pp_rd<0>(w1);
            wf[0][1][0] = pp_rd<2048>(w0); wf[0][1][1] = pp_rd<2048>(w1);
            wf[0][2][0] = pp_rd<4096>(w0); wf[0][2][1] = pp_rd<4096>(w1);
            dma_t(c1, 1, nbo);
            PP_SB(); PP_BARRIER();
            PP_WAIT_W(0);
            PP_SB();
        }
        PP_MSEG(PP_MM(0, 0, 0, 0))
        // ---- L1: w1(G) ; DMA Wb(G+1) ; reads retired BEFORE the barrier (Wa(G+2) is issued in the next slot) -------
        {
            const uint32_t w0 = wa[0] + bo, w1 = wa[1] + bo;
            wf[1][0][0] = pp_rd<6144>(w0); wf[1][0][1] = pp_rd<6144>(w1);
            wf[1][1][0] = pp_rd<8192>(w0); wf[1][1][1] = pp_rd<8192>(w1);
            wf[1][2][0] = pp_rd<10240>(w0); wf[1][2][1] = pp_rd<10240>(w1);
            dma_wb(c1, nbo);
            PP_WAIT_W(1);
            PP_SB(); PP_BARRIER(); PP_SB();
        }
        PP_MSEG(PP_MM(0, 1, 0, 1))
        // ---- L2: t1(G) ; DMA Wa(G+2) ; T0(G+1), T1(G+1) landed --------------------------------------------------------
        {
            const uint32_t t0 = ta[0] + bo, t1 = ta[1] + bo;
            tf[1][0][0] = pp_rd<4096>(t0); tf[1][0][1] = pp_rd<4096>(t1);
            tf[1][1][0] = pp_rd<6144>(t0); tf[1][1][1] = pp_rd<6144>(t1);
            dma_wa(c2, bo);
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            PP_SB(); PP_BARRIER();
            PP_WAIT_T(1);
            PP_SB();
        }
        PP_MSEG(PP_MM(1, 1, 1, 1))
        // ---- L3: t0(G+1) ; DMA T0(G+2) ; W(G+1) landed -------------------------------------------------------------------
        {
            const uint32_t t0 = ta[0] + nbo, t1 = ta[1] + nbo;
            tf[0][0][0] = pp_rd<0>(t0); tf[0][0][1] = pp_rd<0>(t1);
            tf[0][1][0] = pp_rd<2048>(t0); tf[0][1][1] = pp_rd<2048>(t1);
            dma_t(c2, 0, bo);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            PP_SB(); PP_BARRIER();
            PP_WAIT_T(0);
            PP_SB();
        }
        PP_MSEG(PP_MM(1, 0, 1, 0))
        c1 = c2;
        advance(c2);
      }
        // ================= epilogue of tile t_cur (no barrier inside: both groups keep their barrier count) =================
#if PP_ABL & 8
        const uint64_t e0 = __builtin_readcyclecounter();
#endif
        {
            const int64_t mw = (int64_t)(t_cur / ntn) * PP_BM + 64 * wm;          // first row of this wave (wave-uniform); lane row = mw + 16 mt + l15
            const int nb = (t_cur % ntn) * PP_BN + 96 * wn;                        // first column of this wave
            const int64_t left = M - mw;
            const int lim = left > 64 ? 64 : (left < 0 ? 0 : (int)left);           // existing rows below mw
            const char* yb = reinterpret_cast<const char*>(y) + (mw * ldy + nb) * (int64_t)sizeof(TO);
            [[maybe_unused]] const char* y2b = reinterpret_cast<const char*>(y2) + (mw * ldy + nb) * (int64_t)sizeof(TO);
            if constexpr (EPI == PP_F32 || EPI == PP_F32_RES || EPI == PP_F32_RES_SC) {
                // fp32 rows: every residual load of the wave tile is requested before the first add (one round trip, not four)
                [[maybe_unused]] float4 rr[4][6];
                int64_t rowo[4];
                bool okv[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    okv[mt] = l15 + 16 * mt < lim;
                    rowo[mt] = mw + (okv[mt] ? l15 + 16 * mt : (int)(M - 1 - mw));   // masked lanes load an existing row
                }
                if constexpr (EPI != PP_F32) {
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        const float* rp = reinterpret_cast<const float*>(aux) + rowo[mt] * ldaux + nb + 4 * lg;
#pragma unroll
                        for (int nt_ = 0; nt_ < 6; ++nt_) rr[mt][nt_] = load4(rp + 16 * nt_);
                    }
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    [[maybe_unused]] float sc = 1.f;
                    if constexpr (EPI == PP_F32_RES_SC) sc = row_scale[(uint32_t)rowo[mt] / (uint32_t)rps];
                    float* yr = reinterpret_cast<float*>(y) + rowo[mt] * ldy + nb + 4 * lg;
#pragma unroll
                    for (int nt_ = 0; nt_ < 6; ++nt_) {
                        float4 v = make_float4(acc[mt][nt_][0], acc[mt][nt_][1], acc[mt][nt_][2], acc[mt][nt_][3]);
                        if constexpr (EPI == PP_F32_RES_SC) { v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc; }
                        if constexpr (EPI != PP_F32) { v.x += rr[mt][nt_].x; v.y += rr[mt][nt_].y; v.z += rr[mt][nt_].z; v.w += rr[mt][nt_].w; }
                        if (okv[mt]) *reinterpret_cast<float4*>(yr + 16 * nt_) = v;
                    }
                }
            } else {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const bool ok = l15 + 16 * mt < lim;
                    const int64_t rowl = mw + (ok ? l15 + 16 * mt : (int)(M - 1 - mw));       // an existing row for the loads of masked lanes
                    [[maybe_unused]] float sc = 1.f;
                    if constexpr (EPI == PP_DG_PRE || EPI == PP_DG_DER) sc = row_scale ? row_scale[(uint32_t)rowl / (uint32_t)rps] : 1.f;
                    // 16-bit outputs: accumulators nt, nt+1 exchanged by v_permlane16_swap -> every lane holds 8 consecutive
                    // columns: lane group lg -> block nt + (lg & 1), columns 8 (lg >> 1) .. +7
                    [[maybe_unused]] uint4 ax[3];
                    if constexpr (EPI == PP_DG_PRE || EPI == PP_DG_DER) {
                        const bf16_t* ap_ = reinterpret_cast<const bf16_t*>(aux) + rowl * ldaux + nb + 16 * (lg & 1) + 8 * (lg >> 1);
#pragma unroll
                        for (int np = 0; np < 3; ++np) ax[np] = *reinterpret_cast<const uint4*>(ap_ + 32 * np);
                    }
#pragma unroll
                    for (int np = 0; np < 3; ++np) {
                        f32x4 X = acc[mt][2 * np], Y = acc[mt][2 * np + 1];
                        auto pair16 = [&](f32x4 A_, f32x4 B_) {
                            const uint32_t x0 = pack_bf16x2(A_[0], A_[1]), x1 = pack_bf16x2(A_[2], A_[3]);
                            const uint32_t y0 = pack_bf16x2(B_[0], B_[1]), y1 = pack_bf16x2(B_[2], B_[3]);
                            const auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
                            const auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
                            return u32x4{s0[0], s1[0], s0[1], s1[1]};
                        };
                        if constexpr (EPI == PP_DG_PRE || EPI == PP_DG_DER) {
                            // the aux piece is in the stored (exchanged) layout: the same swap hands every lane the values of its own columns
                            const auto u0 = __builtin_amdgcn_permlane16_swap(ax[np].x, ax[np].z, false, false);
                            const auto u1 = __builtin_amdgcn_permlane16_swap(ax[np].y, ax[np].w, false, false);
                            float fx[4] = {lo16_to_f32(u0[0]), hi16_to_f32(u0[0]), lo16_to_f32(u1[0]), hi16_to_f32(u1[0])};
                            float fy[4] = {lo16_to_f32(u0[1]), hi16_to_f32(u0[1]), lo16_to_f32(u1[1]), hi16_to_f32(u1[1])};
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                if constexpr (EPI == PP_DG_PRE) { fx[j] = gelu_grad_fast(fx[j]); fy[j] = gelu_grad_fast(fy[j]); }
                                X[j] *= sc * fx[j];
                                Y[j] *= sc * fy[j];
                            }
                        }
                        if constexpr (EPI == PP_GELU_PRE) st16(y2b, mt, np, lim, pair16(X, Y));
                        if constexpr (EPI == PP_GELU_DER) {
                            f32x4 dX, dY;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                float gv, gd;
                                gelu_and_grad_fast(X[j], gv, gd); X[j] = gv; dX[j] = gd;
                                gelu_and_grad_fast(Y[j], gv, gd); Y[j] = gv; dY[j] = gd;
                            }
                            st16(y2b, mt, np, lim, pair16(dX, dY));
                        }
                        if constexpr (EPI == PP_GELU16 || EPI == PP_GELU_PRE) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) { X[j] = gelu_fast(X[j]); Y[j] = gelu_fast(Y[j]); }
                        }
                        st16(yb, mt, np, lim, pair16(X, Y));
                    }
                }
            }
        }
        t_cur += per;
        if (ti + 1 < my_tiles) init_acc(t_cur);
        PP_SB();
#if PP_ABL & 8
        s_epi += __builtin_readcyclecounter() - e0;
#endif
    }
#if PP_ABL & 8
    if (lane == 0 && wave == 0) {    // whole-loop cycles and 100 MHz ticks -> the clock the loop ran at; K-tiles; epilogue cycles per tile
        float* e = g_pp_stamps + blockIdx.x * 4;
        e[0] = (float)(__builtin_readcyclecounter() - loop_c0); e[1] = (float)(__builtin_readsteadycounter() - loop_r0); e[2] = (float)(my_tiles * nk);
        e[3] = (float)s_epi / my_tiles;
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA may outlive the workgroup
    if (!grp) PP_BARRIER();                               // group 0 is one barrier short of group 1
#undef PP_MM
#undef PP_WAIT_W
#undef PP_WAIT_T
#undef PP_MSEG
}

template <int EPI, typename TO>
static int pp_launch(const void* a, int64_t lda, const void* w, const float* bias, const void* aux, int64_t ldaux,
                     const float* row_scale, int64_t rps, void* y, void* y2, int64_t ldy, int64_t M, int N, int K, hipStream_t st) {
    const int smem = PP_TILES_BYTES + 4 * N;
    static DevInts attr_tab;
    DevInt attr_smem = dev_int(attr_tab);
    if (smem > attr_smem) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_pp_kernel<EPI, TO>), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return MVIT_ELAUNCH;
        attr_smem = smem;
    }
    const int64_t nt = ((M + PP_BM - 1) / PP_BM) * (N / PP_BN);
    const int64_t q = (nt + 7) / 8;
    static DevInts ncu_tab;
    const int ncu = dev_cu_count(ncu_tab);
    if (ncu < 8) return MVIT_ELAUNCH;
    const int per_xcd = ncu / 8;
    const int per = (int)(q < per_xcd ? q : per_xcd);         // one workgroup per CU (32 CUs per XCD on MI355X)
    hipLaunchKernelGGL((linear_pp_kernel<EPI, TO>), dim3((unsigned)(8 * per)), dim3(512), smem, st, (const bf16_t*)a, lda, (const bf16_t*)w,
                       bias, aux, ldaux, row_scale, rps, (TO*)y, (TO*)y2, ldy, M, N, K);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}

// Shapes this kernel takes: N a multiple of 192, K a multiple of 64 (>= 128), 32-bit lane offsets.  MVIT_EUNSUPPORTED otherwise
// (the caller then uses the 128 x 192 kernels of linear.hip).
bool mvit_internal_linear_pp_ok(int64_t lda, int64_t M, int N, int K) {
    return N % PP_BN == 0 && K % PP_BK == 0 && K >= 2 * PP_BK && (lda & 7) == 0 && 512 * lda < (1ll << 31) && (int64_t)N * K < (1ll << 30) &&
           N <= 8192 && M < (1ll << 31) && ((M + PP_BM - 1) / PP_BM) * (N / PP_BN) < (1ll << 30);
}

int mvit_internal_linear_pp(int epi, const void* a, int64_t lda, const void* w, const float* bias, const void* aux, int64_t ldaux,
                            const float* row_scale, int64_t rps, void* y, void* y2, int64_t ldy, int64_t M, int N, int K, hipStream_t st) {
    if (!mvit_internal_linear_pp_ok(lda, M, N, K)) return MVIT_EUNSUPPORTED;
#define PPL(E, T) case E: return pp_launch<E, T>(a, lda, w, bias, aux, ldaux, row_scale, rps, y, y2, ldy, M, N, K, st)
    switch (epi) {
        PPL(PP_B16, bf16_t);
        PPL(PP_GELU16, bf16_t);
        PPL(PP_GELU_PRE, bf16_t);
        PPL(PP_GELU_DER, bf16_t);
        PPL(PP_F32, float);
        PPL(PP_F32_RES, float);
        PPL(PP_F32_RES_SC, float);
        PPL(PP_DG_PRE, bf16_t);
        PPL(PP_DG_DER, bf16_t);
    }
#undef PPL
    return MVIT_EINVAL;
}
