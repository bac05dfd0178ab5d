// K = 96 linear layers on long token streams (block 0 / block 1 of the model: qkv 96 -> 288 / 576, fc1 96 -> 384; 0.8 M rows at
// B = 8 @448): y = epilogue(a . w^T + bias), 16-bit operands and outputs, fp32 accumulate.   reference: nn.Linear of
// slowfast/models/attention.py:231 (qkv) and common.py:27-31 (fc1 + GELU).
//
// These layers move 4-6x more bytes out than in and do 96 multiply-adds per output: they are HBM-bound (617 MB written for fc1 at
// 5.2 TB/s = 119 us), but the general 128 x 192 kernels ran them at 2.6-3.2 TB/s (profiles/r3_thin_gemm.txt) -- per 128 x 192 tile they
// re-load the 192 x 96 weight panel AND walk K as two 64-wide slabs (the second half empty): 80 KB of LDS-DMA per 49 KB of output, and
// the LDS-DMA path of a CU moves ~11 B/clk (170 us for fc1 by itself).  Here:
//   * the whole weight matrix (N x 96, <= 108 KiB) is loaded into LDS ONCE per workgroup and stays; one persistent workgroup per CU
//     streams 128-row token tiles through a double buffer: 24 KB of DMA per tile, each token row read from memory once for all N columns;
//   * a wave owns 64 rows x 96 columns (six 32 x 32 accumulators, product computed transposed so a lane owns output rows and leaves
//     16-byte pieces: the epilogue of linear.hip); N = 288 / 384 run 6 / 8 waves in one pass, N = 576 six waves in two passes over the
//     column halves from the same token tile;
//   * LDS images are [rows][192 B] with the rotation swizzle of the attention kernels (chunk + ((row >> 2) & 3) mod 12: conflict-free
//     for the 32-row ds_read_b128 fragment reads), written lane-linear by global_load_lds_dwordx4 with the swizzle on the source address;
//   * one s_barrier per tile; the next tile's DMA is issued right behind it and the output stores trail (counted vmcnt).
// Epilogues: 16-bit y = acc | GELU(acc) | GELU(acc) with y2 = acc (training, pre-activation kept) | with y2 = GELU'(acc).
#include <stdlib.h>

#include "common.h"

#define K9_ROWB 192                 // bytes per image row (96 x 16 bit)
#define K9_TILE_M 128
#define K9_ABYTES (K9_TILE_M * K9_ROWB)      // 24 KiB

enum { K9_B16 = 0, K9_GELU16 = 1, K9_GELU_PRE = 2, K9_GELU_DER = 3 };

__device__ __forceinline__ void k9_dma(const char* base, uint32_t off, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(off), "s"(base) : "memory");
}

// NWP: 96-column blocks per pass = waves along N (3 or 4); the workgroup has 2 * NWP waves.  PASSES: column passes per token tile.
template <int NWP, int PASSES, int EPI>
__global__ __launch_bounds__(128 * NWP) void linear_k96_kernel(const bf16_t* __restrict__ a, int64_t lda, const bf16_t* __restrict__ w,
                                                               const float* __restrict__ bias, bf16_t* __restrict__ y,
                                                               bf16_t* __restrict__ y2, int64_t ldy, int64_t M) {
    constexpr int NW = 2 * NWP;                     // waves
    constexpr int N = 96 * NWP * PASSES;
    constexpr int PA = 24 / NW;                     // token-tile DMA pieces per wave (24 pieces of 1 KiB)
    constexpr int PW = (N * 12 / 64) / NW;          // weight-image pieces per wave
    constexpr int NST = 12 * PASSES * (EPI >= K9_GELU_PRE ? 2 : 1);     // vector stores per wave and tile
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [W image N x 192][A0][A1]
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wm = wave / NWP, wn = wave - wm * NWP;
    const int r = lane & 31, h = lane >> 5;
    const int64_t ntiles = (M + K9_TILE_M - 1) / K9_TILE_M;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t lds_a = lds0 + N * K9_ROWB;

    // ---- weights: once -------------------------------------------------------------------------------------------------------
    // LDS chunk position p = 64 * piece + lane of an image holds source chunk (p % 12 - rot(row)) mod 12 of row p / 12
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int p = 64 * (PW * wave + i) + lane;
        const int row = p / 12, pos = p - row * 12;
        int c = pos - ((row >> 2) & 3);
        c = c < 0 ? c + 12 : c;
        k9_dma(reinterpret_cast<const char*>(w), (uint32_t)(row * K9_ROWB + c * 16), lds0 + 1024 * (PW * wave + i));
    }
    // ---- token tiles: per-lane source offsets of this wave's pieces (full tile; the ragged last tile recomputes them) ------------
    int a_row[PA], a_c16[PA];
    uint32_t a_off[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int p = 64 * (PA * wave + i) + lane;
        const int row = p / 12, pos = p - row * 12;
        int c = pos - ((row >> 2) & 3);
        c = c < 0 ? c + 12 : c;
        a_row[i] = row;
        a_c16[i] = c * 16;
        a_off[i] = (uint32_t)(row * (int)lda * 2 + c * 16);
    }
    auto dma_tile = [&](int64_t t, int buf) {
        const int64_t m0 = t * K9_TILE_M;
        const char* base = reinterpret_cast<const char*>(a + m0 * lda);      // wave-uniform
        const uint32_t dst = lds_a + buf * K9_ABYTES + 1024 * (PA * wave);
        if (m0 + K9_TILE_M <= M) {
#pragma unroll
            for (int i = 0; i < PA; ++i) k9_dma(base, a_off[i], dst + 1024 * i);
        } else {      // rows past M re-read the last row (never stored)
            const int last = (int)(M - 1 - m0);
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                const int row = a_row[i] < last ? a_row[i] : last;
                k9_dma(base, (uint32_t)(row * (int)lda * 2 + a_c16[i]), dst + 1024 * i);
            }
        }
    };
    int64_t t = blockIdx.x;
    if (t < ntiles) dma_tile(t, 0);

    // fragment addressing: chunk 2 ks + h of a row, rotated
    int roff[6];
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
        int p = 2 * ks + h + ((r >> 2) & 3);
        p = p >= 12 ? p - 12 : p;
        roff[ks] = p * 16;
    }
    const char* xrow = smem + N * K9_ROWB + (64 * wm + r) * K9_ROWB;       // + buf * K9_ABYTES + 32 mb rows
    const char* wrow = smem + (96 * wn + r) * K9_ROWB;                       // + pass * 96 NWP rows + 32 nb rows

    int buf = 0;
    bool first = true;
    for (; t < ntiles; t += gridDim.x) {
        const bool ragged = (t + 1) * K9_TILE_M > M;
        // tile t (and, the first time, the weights) has landed: its DMA was issued before the previous tile's NST stores
        if (first || ragged) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
        first = false;
        __builtin_amdgcn_s_barrier();                 // ... for every wave; everyone is done reading the other buffer
        if (t + gridDim.x < ntiles) dma_tile(t + gridDim.x, buf ^ 1);
        const char* xr = xrow + buf * K9_ABYTES;
        const int64_t mw = t * K9_TILE_M + 64 * wm;
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int nw0 = 96 * (NWP * ps + wn);                             // first column of this wave in this pass
            f32x16 acc[2][3];
#pragma unroll
            for (int nb = 0; nb < 3; ++nb) {      // the bias is the accumulators' initial value (wave-uniform loads)
                const float4* bp = reinterpret_cast<const float4*>(bias + nw0 + 32 * nb);
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const float4 lo = bp[2 * g4], hi = bp[2 * g4 + 1];
                    const float b0 = h ? hi.x : lo.x, b1 = h ? hi.y : lo.y, b2 = h ? hi.z : lo.z, b3 = h ? hi.w : lo.w;
                    acc[0][nb][4 * g4 + 0] = b0; acc[1][nb][4 * g4 + 0] = b0;
                    acc[0][nb][4 * g4 + 1] = b1; acc[1][nb][4 * g4 + 1] = b1;
                    acc[0][nb][4 * g4 + 2] = b2; acc[1][nb][4 * g4 + 2] = b2;
                    acc[0][nb][4 * g4 + 3] = b3; acc[1][nb][4 * g4 + 3] = b3;
                }
            }
            const char* wr = wrow + ps * (96 * NWP * K9_ROWB);
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) {
                bf16x8 xf[2], wf[3];
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) xf[mb] = *reinterpret_cast<const bf16x8*>(xr + 32 * mb * K9_ROWB + roff[ks]);
#pragma unroll
                for (int nb = 0; nb < 3; ++nb) wf[nb] = *reinterpret_cast<const bf16x8*>(wr + 32 * nb * K9_ROWB + roff[ks]);
#pragma unroll
                for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) acc[mb][nb] = mfma16(wf[nb], xf[mb], acc[mb][nb]);
            }
            // ---- epilogue from registers: lane = output row, quads of 4 consecutive columns; 16-byte pieces via permlane32_swap --
            // Two outputs: ALL stores of y first, then all of y2 (the packed y2 pieces wait in 48 registers).  In a pure store stream,
            // alternating between two matrices store by store costs a third of the write bandwidth (3.5 against 5.1 TB/s:
            // tools/probes/store_bw, patterns 3 / 5); here the ordered form runs a steady 360 us where the alternating one ran
            // 341 ... 430 (the same reordering in the ping-pong kernel's epilogue, whose stores are spread out anyway, was 6 % slower).
            [[maybe_unused]] uint4 y2p[2][3][2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const int64_t m = mw + 32 * mb + r;
                const bool ok = !ragged || m < M;
#pragma unroll
                for (int nb = 0; nb < 3; ++nb) {
                    float4 v[4];
                    [[maybe_unused]] float4 u[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v[q] = make_float4(acc[mb][nb][4 * q], acc[mb][nb][4 * q + 1], acc[mb][nb][4 * q + 2], acc[mb][nb][4 * q + 3]);
                        if constexpr (EPI == K9_GELU16) {
                            v[q].x = gelu_fast(v[q].x); v[q].y = gelu_fast(v[q].y); v[q].z = gelu_fast(v[q].z); v[q].w = gelu_fast(v[q].w);
                        } else if constexpr (EPI == K9_GELU_PRE) {
                            u[q] = v[q];
                            v[q].x = gelu_fast(v[q].x); v[q].y = gelu_fast(v[q].y); v[q].z = gelu_fast(v[q].z); v[q].w = gelu_fast(v[q].w);
                        } else if constexpr (EPI == K9_GELU_DER) {
                            gelu_and_grad_fast(v[q].x, v[q].x, u[q].x); gelu_and_grad_fast(v[q].y, v[q].y, u[q].y);
                            gelu_and_grad_fast(v[q].z, v[q].z, u[q].z); gelu_and_grad_fast(v[q].w, v[q].w, u[q].w);
                        }
                    }
                    const int64_t o = m * ldy + nw0 + 32 * nb;
#pragma unroll
                    for (int q = 0; q < 4; q += 2) {
                        // lane (r,0) holds cols 8q..8q+3, lane (r,1) 8q+4..8q+7 of one row: after the swaps the lower half holds 8 consecutive
                        // columns of quad-pair q, the upper half those of quad-pair q+1
                        {
                            const uint32_t a0 = pack_bf16x2(v[q].x, v[q].y), a1 = pack_bf16x2(v[q].z, v[q].w);
                            const uint32_t b0 = pack_bf16x2(v[q + 1].x, v[q + 1].y), b1 = pack_bf16x2(v[q + 1].z, v[q + 1].w);
                            const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                            const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                            if (ok) *reinterpret_cast<uint4*>(y + o + 8 * (q + h)) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                        }
                        if constexpr (EPI >= K9_GELU_PRE) {
                            const uint32_t a0 = pack_bf16x2(u[q].x, u[q].y), a1 = pack_bf16x2(u[q].z, u[q].w);
                            const uint32_t b0 = pack_bf16x2(u[q + 1].x, u[q + 1].y), b1 = pack_bf16x2(u[q + 1].z, u[q + 1].w);
                            const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                            const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                            y2p[mb][nb][q >> 1] = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                        }
                    }
                }
            }
            if constexpr (EPI >= K9_GELU_PRE) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    const int64_t m = mw + 32 * mb + r;
                    const bool ok = !ragged || m < M;
#pragma unroll
                    for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                        for (int q = 0; q < 4; q += 2)
                            if (ok) *reinterpret_cast<uint4*>(y2 + m * ldy + nw0 + 32 * nb + 8 * (q + h)) = y2p[mb][nb][q >> 1];
                }
            }
        }
        buf ^= 1;
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------
bool mvit_internal_linear_k96_ok(int64_t lda, int64_t M, int N, int K) {
    static const bool on = !(getenv("MVIT_GEMM_K96") && atoi(getenv("MVIT_GEMM_K96")) == 0);
    return on && K == 96 && (N == 288 || N == 384 || N == 576) && M >= 32768 && (lda & 7) == 0 && 128 * lda * 2 < (1ll << 31);
}

template <int NWP, int PASSES, int EPI>
static int k96_launch(const void* a, int64_t lda, const void* w, const float* bias, void* y, void* y2, int64_t ldy, int64_t M,
                      hipStream_t st) {
    constexpr int N = 96 * NWP * PASSES;
    constexpr int SMEM = N * K9_ROWB + 2 * K9_ABYTES;
    static DevInts ncu_tab;
    const int ncu = dev_cu_count(ncu_tab);
    if (ncu <= 0) return MVIT_ELAUNCH;
    static DevFlags attr_tab;
    DevFlag attr_done = dev_flag(attr_tab);
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_k96_kernel<NWP, PASSES, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                SMEM) != hipSuccess)
            return MVIT_ELAUNCH;
        attr_done = true;
    }
    const int64_t ntiles = (M + K9_TILE_M - 1) / K9_TILE_M;
    const unsigned grid = (unsigned)(ntiles < ncu ? ntiles : ncu);
    hipLaunchKernelGGL((linear_k96_kernel<NWP, PASSES, EPI>), dim3(grid), dim3(128 * NWP), SMEM, st, (const bf16_t*)a, lda, (const bf16_t*)w, bias,
                       (bf16_t*)y, (bf16_t*)y2, ldy, M);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}

// epi: 0 y = acc + bias, 1 y = GELU, 2 y = GELU and y2 = pre-activation, 3 y = GELU and y2 = GELU'.  bias required.
int mvit_internal_linear_k96(int epi, const void* a, int64_t lda, const void* w, const float* bias, void* y, void* y2, int64_t ldy, int64_t M,
                             int N, hipStream_t st) {
    if (!bias || (ldy & 7) || (epi >= K9_GELU_PRE && !y2)) return MVIT_EINVAL;
#define K9_N(NWP, PASSES)                                                                                   \
    switch (epi) {                                                                                          \
        case K9_B16: return k96_launch<NWP, PASSES, K9_B16>(a, lda, w, bias, y, y2, ldy, M, st);            \
        case K9_GELU16: return k96_launch<NWP, PASSES, K9_GELU16>(a, lda, w, bias, y, y2, ldy, M, st);      \
        case K9_GELU_PRE: return k96_launch<NWP, PASSES, K9_GELU_PRE>(a, lda, w, bias, y, y2, ldy, M, st);  \
        case K9_GELU_DER: return k96_launch<NWP, PASSES, K9_GELU_DER>(a, lda, w, bias, y, y2, ldy, M, st);  \
        default: return MVIT_EINVAL;                                                                        \
    }
    if (N == 288) { K9_N(3, 1) }
    if (N == 384) { K9_N(4, 1) }
    if (N == 576) { K9_N(3, 2) }
#undef K9_N
    return MVIT_EUNSUPPORTED;
}
