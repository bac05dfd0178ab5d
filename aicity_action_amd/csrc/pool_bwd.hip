// Backward of the pooling conv + LayerNorm (attention_pool, conv variant).  v1: straightforward gather kernels
// (HBM/L2-bound, fp32 accumulate), three launches:
//   1. pool_ln_bwd_kernel : recompute conv + LN statistics per output token, LN backward -> d_conv (act-typed),
//                           partial sums of d_gamma / d_beta (deterministic two-stage reduction)
//   2. pool_dgrad_kernel  : transposed depthwise conv of d_conv, written into the q/k/v slice of the fused d_qkv buffer
//   3. pool_wgrad_kernel  : dw[c][tap] += sum_tokens d_conv[tok][c] * in[pos(tok,tap)][c]   (fp32 atomics, 27x96 outputs)
#include "common.h"

#define PB_MAXBLK 256
#define PW_MAXBLK 512

extern "C" int mvit_colsum(const void* a, int a_dtype, int64_t M, int N, const float* row_scale, int64_t rows_per_scale,
                           float* out, int accumulate, float* workspace, void* stream);
int mvit_internal_pool_dgrad_tiled(const void* dconv, const float* w, void* dqkv, int64_t ld, int chan_off, int B, int heads, int T,
                                   int H, int W, int act_dtype, hipStream_t st);
int mvit_internal_pool_dgrad2_tiled(const void* dconv, const float* w, void* dqkv, int64_t ld, int chan_off, int B, int heads, int T,
                                    int H, int W, int act_dtype, hipStream_t st);
int mvit_internal_pool_wgrad_tiled(const void* qkv, int64_t ld, int chan_off, const void* dconv, float* part, int B, int heads,
                                   int T, int H, int W, int stride_hw, int act_dtype, hipStream_t st);
int mvit_internal_pool_dgrad2_tiled_kv(const void* dconv_kv, const float* w_k, const float* w_v, void* dqkv, int64_t ld, int chan_off_k,
                                       int B, int heads, int T, int H, int W, int act_dtype, hipStream_t st);
int mvit_internal_pool_wgrad_march(const void* qkv, int64_t ld, int chan_off, const void* dconv, float* part, int B, int heads, int T, int H,
                                   int W, int stride_hw, int nset, hipStream_t st, const void* xhat, const void* dout, const float* rstd,
                                   const float* gamma, const float* gamma2, float* part_ln);      // pool_march.hip (xhat != null: LayerNorm backward fused in front)
int mvit_internal_pool_wgrad_tiled_kv(const void* qkv, int64_t ld, int chan_off_k, const void* dconv_kv, float* part, int B, int heads,
                                      int T, int H, int W, int act_dtype, hipStream_t st);
int mvit_internal_pool_ln_bwd_tiled(const void* qkv, int64_t ld, int chan_off, const float* w, const float* gamma, const void* dout,
                                    void* dconv, float* part, int B, int heads, int T, int H, int W, int stride_hw, float eps,
                                    int act_dtype, hipStream_t st);

template <typename TA>
__global__ __launch_bounds__(256) void pool_ln_bwd_kernel(const TA* __restrict__ qkv, int64_t ld, int chan_off,
                                                          const float* __restrict__ w, const float* __restrict__ gamma,
                                                          const TA* __restrict__ dout, TA* __restrict__ dconv,
                                                          float* __restrict__ part, int B, int heads, int T, int H, int W,
                                                          int Ho, int Wo, int s, float eps) {
    constexpr int CW = 16 / sizeof(TA);
    constexpr int NCH = 24 / CW;
    __shared__ __attribute__((aligned(16))) float wsm[27 * 96];
    __shared__ float red[64][4];   // reused for the final reduction in chunks
    for (int i = threadIdx.x; i < 27 * 96; i += 256) {
        const int tap = i / 96, c = i - tap * 96;
        wsm[i] = w[c * 27 + tap];
    }
    __syncthreads();
    const int j = threadIdx.x & 3;
    float g[24], ag[24], ab[24];
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int e = 0; e < CW; ++e) g[i * CW + e] = gamma[CW * (j + 4 * i) + e];
#pragma unroll
    for (int e = 0; e < 24; ++e) { ag[e] = 0.f; ab[e] = 0.f; }
    const int64_t Lout = (int64_t)T * Ho * Wo;
    const int64_t total = (int64_t)B * heads * Lout;
    const int64_t Nin = (int64_t)T * H * W;
    for (int64_t it0 = (int64_t)blockIdx.x * 64; it0 < total; it0 += (int64_t)gridDim.x * 64) {
        const int64_t it = it0 + (threadIdx.x >> 2);
        const bool ok = it < total;
        const int64_t itc = ok ? it : total - 1;
        int64_t rem = itc;
        const int xo = (int)(rem % Wo); rem /= Wo;
        const int yo = (int)(rem % Ho); rem /= Ho;
        const int to = (int)(rem % T); rem /= T;
        const int gh = (int)(rem % heads);
        const int b = (int)(rem / heads);
        const TA* base = qkv + (int64_t)b * Nin * ld + chan_off + gh * 96;
        float acc[24];
#pragma unroll
        for (int e = 0; e < 24; ++e) acc[e] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
            const int ti = to + dt - 1;
            if (ti < 0 || ti >= T) continue;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int yi = yo * s + dy - 1;
                if (yi < 0 || yi >= H) continue;
                // the three dx taps of this row: 16-byte loads issued together at clamped x, out-of-frame taps zeroed afterwards
                const TA* rowp = base + (((int64_t)ti * H + yi) * W) * ld;
                float xv[3][NCH * CW];
                float msk[3];
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int xi0 = xo * s + dx - 1;
                    const int xi = xi0 < 0 ? 0 : (xi0 >= W ? W - 1 : xi0);
                    msk[dx] = xi0 == xi ? 1.f : 0.f;
                    const TA* p = rowp + (int64_t)xi * ld;
#pragma unroll
                    for (int i = 0; i < NCH; ++i) {
                        const int c0 = CW * (j + 4 * i);
                        if constexpr (sizeof(TA) == 2) {
                            float4 lo, hi;
                            load8(p + c0, lo, hi);
                            xv[dx][i * 8 + 0] = lo.x; xv[dx][i * 8 + 1] = lo.y; xv[dx][i * 8 + 2] = lo.z; xv[dx][i * 8 + 3] = lo.w;
                            xv[dx][i * 8 + 4] = hi.x; xv[dx][i * 8 + 5] = hi.y; xv[dx][i * 8 + 6] = hi.z; xv[dx][i * 8 + 7] = hi.w;
                        } else {
                            const float4 v = load4(p + c0);
                            xv[dx][i * 4 + 0] = v.x; xv[dx][i * 4 + 1] = v.y; xv[dx][i * 4 + 2] = v.z; xv[dx][i * 4 + 3] = v.w;
                        }
                    }
                }
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float* wt = wsm + ((dt * 3 + dy) * 3 + dx) * 96;
#pragma unroll
                    for (int i = 0; i < NCH; ++i) {
                        const int c0 = CW * (j + 4 * i);
#pragma unroll
                        for (int e = 0; e < CW; e += 4) {
                            const float4 ww = *reinterpret_cast<const float4*>(wt + c0 + e);
                            acc[i * CW + e] = fmaf(xv[dx][i * CW + e] * msk[dx], ww.x, acc[i * CW + e]);
                            acc[i * CW + e + 1] = fmaf(xv[dx][i * CW + e + 1] * msk[dx], ww.y, acc[i * CW + e + 1]);
                            acc[i * CW + e + 2] = fmaf(xv[dx][i * CW + e + 2] * msk[dx], ww.z, acc[i * CW + e + 2]);
                            acc[i * CW + e + 3] = fmaf(xv[dx][i * CW + e + 3] * msk[dx], ww.w, acc[i * CW + e + 3]);
                        }
                    }
                }
            }
        }
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < 24; ++e) sum += acc[e];
        sum += __shfl_xor(sum, 1, 64);
        sum += __shfl_xor(sum, 2, 64);
        const float mean = sum * (1.0f / 96.0f);
        float sq = 0.f;
#pragma unroll
        for (int e = 0; e < 24; ++e) { acc[e] -= mean; sq += acc[e] * acc[e]; }
        sq += __shfl_xor(sq, 1, 64);
        sq += __shfl_xor(sq, 2, 64);
        const float rstd = 1.0f / sqrtf(sq * (1.0f / 96.0f) + eps);
        float dyv[24];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int e = 0; e < CW; e += 4) {
                float4 d4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) d4 = load4(dout + it * 96 + CW * (j + 4 * i) + e);
                dyv[i * CW + e] = d4.x; dyv[i * CW + e + 1] = d4.y; dyv[i * CW + e + 2] = d4.z; dyv[i * CW + e + 3] = d4.w;
            }
#pragma unroll
        for (int e = 0; e < 24; ++e) {
            acc[e] *= rstd;                  // xhat
            ag[e] += dyv[e] * acc[e];
            ab[e] += dyv[e];
            dyv[e] *= g[e];                  // gamma * dy
            c1 += dyv[e];
            c2 += dyv[e] * acc[e];
        }
        c1 += __shfl_xor(c1, 1, 64); c1 += __shfl_xor(c1, 2, 64);
        c2 += __shfl_xor(c2, 1, 64); c2 += __shfl_xor(c2, 2, 64);
        c1 *= (1.0f / 96.0f);
        c2 *= (1.0f / 96.0f);
        if (ok) {
            TA* o = dconv + it * 96;
#pragma unroll
            for (int i = 0; i < NCH; ++i)
#pragma unroll
                for (int e = 0; e < CW; e += 4) {
                    const int k = i * CW + e;
                    float4 v;
                    v.x = rstd * (dyv[k] - c1 - acc[k] * c2);
                    v.y = rstd * (dyv[k + 1] - c1 - acc[k + 1] * c2);
                    v.z = rstd * (dyv[k + 2] - c1 - acc[k + 2] * c2);
                    v.w = rstd * (dyv[k + 3] - c1 - acc[k + 3] * c2);
                    store4(o + CW * (j + 4 * i) + e, v);
                }
        }
    }
    // block reduction of the dgamma / dbeta partials: lanes with equal j own the same 24 channels
    auto block_reduce = [&](const float (&src)[24], int pass) {
#pragma unroll
        for (int e = 0; e < 24; ++e) {
            __syncthreads();
            red[threadIdx.x >> 2][j] = src[e];
            __syncthreads();
            if (threadIdx.x < 4) {
                float t = 0.f;
                for (int rr = 0; rr < 64; ++rr) t += red[rr][threadIdx.x];
                // channel of (j = threadIdx.x, element e): chunk i = e / CW, offset e % CW
                const int c = CW * (threadIdx.x + 4 * (e / CW)) + (e % CW);
                part[(int64_t)blockIdx.x * 192 + pass * 96 + c] = t;
            }
        }
    };
    block_reduce(ag, 0);
    block_reduce(ab, 1);
}

// out[j] (+)= sum_b part[b][j], width columns; columns [0,split) -> out_a, rest -> out_b: the library's deterministic
// single-launch column reduce (backward_rowops.hip: row slices parked in scratch, the last-arriving slice adds them in order)
static int launch_pool_reduce(const float* part, int nparts, int width, float* out_a, float* out_b, int split, int accumulate,
                              hipStream_t st, int defer_ok = 0) {
    return mvit_internal_reduce_partials(part, nparts, width, out_a, out_b, split, accumulate, st, defer_ok);
}

template <typename TA>
__global__ __launch_bounds__(256) void pool_dgrad_kernel(const TA* __restrict__ dconv, const float* __restrict__ w,
                                                         TA* __restrict__ dqkv, int64_t ld, int chan_off, int B, int heads,
                                                         int T, int H, int W, int Ho, int Wo, int s) {
    constexpr int CW = 16 / sizeof(TA);
    constexpr int NCH = 24 / CW;
    __shared__ __attribute__((aligned(16))) float wsm[27 * 96];
    for (int i = threadIdx.x; i < 27 * 96; i += 256) {
        const int tap = i / 96, c = i - tap * 96;
        wsm[i] = w[c * 27 + tap];
    }
    __syncthreads();
    const int j = threadIdx.x & 3;
    const int64_t Nin = (int64_t)T * H * W;
    const int64_t total = (int64_t)B * heads * Nin;
    const int64_t Lout = (int64_t)T * Ho * Wo;
    for (int64_t it0 = (int64_t)blockIdx.x * 64; it0 < total; it0 += (int64_t)gridDim.x * 64) {
        const int64_t it = it0 + (threadIdx.x >> 2);
        if (it >= total) continue;
        int64_t rem = it;
        int x = (int)(rem % W); rem /= W;
        // strided pooling: walk x class by class (x mod s) so that the 16 tokens of a wave share their valid tap set -- the tap
        // loop's parity tests become wave-uniform branches instead of 27 masked iterations
        if (s > 1 && W % s == 0) {
            const int nxs = W / s;
            x = (x % nxs) * s + x / nxs;
        }
        const int y = (int)(rem % H); rem /= H;
        const int t = (int)(rem % T); rem /= T;
        const int gh = (int)(rem % heads);
        const int b = (int)(rem / heads);
        float acc[24];
#pragma unroll
        for (int e = 0; e < 24; ++e) acc[e] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
            const int to = t + 1 - dt;
            if (to < 0 || to >= T) continue;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int ny = y + 1 - dy;
                if (ny < 0 || ny % s) continue;
                const int yo = ny / s;
                if (yo >= Ho) continue;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int nx = x + 1 - dx;
                    if (nx < 0 || nx % s) continue;
                    const int xo = nx / s;
                    if (xo >= Wo) continue;
                    const TA* p = dconv + (((int64_t)(b * heads + gh)) * Lout + ((int64_t)to * Ho + yo) * Wo + xo) * 96;
                    const float* wt = wsm + ((dt * 3 + dy) * 3 + dx) * 96;
#pragma unroll
                    for (int i = 0; i < NCH; ++i) {
                        const int c0 = CW * (j + 4 * i);
#pragma unroll
                        for (int e = 0; e < CW; e += 4) {
                            const float4 v = load4(p + c0 + e);
                            const float4 ww = *reinterpret_cast<const float4*>(wt + c0 + e);
                            acc[i * CW + e] = fmaf(v.x, ww.x, acc[i * CW + e]);
                            acc[i * CW + e + 1] = fmaf(v.y, ww.y, acc[i * CW + e + 1]);
                            acc[i * CW + e + 2] = fmaf(v.z, ww.z, acc[i * CW + e + 2]);
                            acc[i * CW + e + 3] = fmaf(v.w, ww.w, acc[i * CW + e + 3]);
                        }
                    }
                }
            }
        }
        TA* o = dqkv + ((int64_t)b * Nin + ((int64_t)t * H + y) * W + x) * ld + chan_off + gh * 96;
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int e = 0; e < CW; e += 4)
                store4(o + CW * (j + 4 * i) + e, make_float4(acc[i * CW + e], acc[i * CW + e + 1], acc[i * CW + e + 2], acc[i * CW + e + 3]));
    }
}

// dw partials: thread = (16-byte channel chunk, tap) -> 2 wide loads per 8 (bf16) / 4 (fp32) FMAs; 3 token lanes per block walk
// every 3rd token of the block's contiguous token range; LDS combine; one partial row [2592] per block.
template <typename TA>
__global__ __launch_bounds__(1024) void pool_wgrad_kernel(const TA* __restrict__ qkv, int64_t ld, int chan_off,
                                                          const TA* __restrict__ dconv, float* __restrict__ part, int B,
                                                          int heads, int T, int H, int W, int Ho, int Wo, int s) {
    constexpr int CW = 16 / sizeof(TA);
    constexpr int NCG = 96 / CW;             // channel groups (12 bf16 / 24 fp32)
    constexpr int NPAIR = NCG * 27;          // (cg, tap) pairs: 324 / 648
    constexpr int TL = 1024 / NPAIR;         // token lanes: 3 / 1
    __shared__ float red[(TL > 1 ? TL - 1 : 1)][2592];
    const int tl = threadIdx.x / NPAIR, pr = threadIdx.x % NPAIR;
    const bool on = tl < TL;
    const int cg = pr % NCG, tap = pr / NCG;
    const int dt = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
    const int64_t Lout = (int64_t)T * Ho * Wo;
    const int64_t total = (int64_t)B * heads * Lout;
    const int64_t Nin = (int64_t)T * H * W;
    float acc[CW];
#pragma unroll
    for (int e = 0; e < CW; ++e) acc[e] = 0.f;
    const int64_t per = (total + gridDim.x - 1) / gridDim.x;
    const int64_t beg = (int64_t)blockIdx.x * per;
    const int64_t end = beg + per < total ? beg + per : total;
    if (on)
        for (int64_t it = beg + tl; it < end; it += TL) {
            int64_t rem = it;
            const int xo = (int)(rem % Wo); rem /= Wo;
            const int yo = (int)(rem % Ho); rem /= Ho;
            const int to = (int)(rem % T); rem /= T;
            const int gh = (int)(rem % heads);
            const int b = (int)(rem / heads);
            const int ti = to + dt - 1, yi = yo * s + dy - 1, xi = xo * s + dx - 1;
            if (ti < 0 || ti >= T || yi < 0 || yi >= H || xi < 0 || xi >= W) continue;
            const TA* ip = qkv + ((int64_t)b * Nin + ((int64_t)ti * H + yi) * W + xi) * ld + chan_off + gh * 96 + CW * cg;
            const TA* dp = dconv + it * 96 + CW * cg;
#pragma unroll
            for (int e = 0; e < CW; e += 4) {
                const float4 d4 = load4(dp + e), v4 = load4(ip + e);
                acc[e] = fmaf(d4.x, v4.x, acc[e]); acc[e + 1] = fmaf(d4.y, v4.y, acc[e + 1]);
                acc[e + 2] = fmaf(d4.z, v4.z, acc[e + 2]); acc[e + 3] = fmaf(d4.w, v4.w, acc[e + 3]);
            }
        }
    if (on && tl > 0) {
#pragma unroll
        for (int e = 0; e < CW; ++e) red[tl - 1][(CW * cg + e) * 27 + tap] = acc[e];
    }
    __syncthreads();
    if (tl == 0) {
#pragma unroll
        for (int e = 0; e < CW; ++e) {
            const int o = (CW * cg + e) * 27 + tap;
            float t = acc[e];
#pragma unroll
            for (int k = 0; k < TL - 1; ++k) t += red[k][o];
            part[(int64_t)blockIdx.x * 2592 + o] = t;
        }
    }
}

// Workspace layout of the pooling-conv backward, ONE formula for the size query and for the entry points (round 5: the two had their own
// copies, and a change to one of them overran the allocation): [prow][192] LayerNorm partial rows | [wrows][2592] weight-gradient rows.
//   prow : one row per workgroup of whichever form runs -- the row-wise LayerNorm backward (<= 2048 blocks, clamped to prow), the 8-wide
//          tiled form (exact-fp32 path), the fused LayerNorm-backward + weight-gradient march kernel (one per 7 x 7 tile)
//   wrows: one row per workgroup of the weight-gradient kernel -- generic (<= PW_MAXBLK), 8-wide tiles, 7 x 7 tiles
static void pool_bwd_rows(int B, int heads, int H, int W, int stride_hw, int64_t& prow, int64_t& wrows) {
    const int Ho = (H - 1) / stride_hw + 1, Wo = (W - 1) / stride_hw + 1;
    prow = PB_MAXBLK;
    if (stride_hw == 1) prow = (int64_t)((Wo + 7) / 8) * ((Ho + 7) / 8) * B * heads;
    if (stride_hw == 2) prow = (int64_t)((Wo + 7) / 8) * ((Ho + 3) / 4) * B * heads;
    int64_t mrows = 0;
    if (stride_hw == 1 || stride_hw == 2) mrows = (int64_t)((Wo + 6) / 7) * ((Ho + 6) / 7) * B * heads;
    if (mrows > prow) prow = mrows;
    if (prow < PB_MAXBLK) prow = PB_MAXBLK;
    wrows = prow > PW_MAXBLK ? prow : PW_MAXBLK;
}
extern "C" int64_t mvit_pool_bwd_workspace_bytes(int B, int heads, int T, int H, int W, int stride_hw) {
    (void)T;
    int64_t prow, wrows;
    pool_bwd_rows(B, heads, H, W, stride_hw, prow, wrows);
    return (prow * 192 + wrows * 2592) * (int64_t)sizeof(float);
}

// Data gradient for strides >= 3: the 3x3 spatial footprints of neighbouring outputs do not overlap, so an input token (t, y, x)
// receives from at most ONE spatial tap (dy = (y+1) mod s when that is < 3, output row (y+1)/s; same in x) and three temporal
// ones; rows / columns with (y+1) mod s >= 3 get zero.  All 18 loads of a token are requested at clamped positions before the
// arithmetic, masked by a 0/1 factor.
template <typename TA>
__global__ __launch_bounds__(256) void pool_dgrad_sparse_kernel(const TA* __restrict__ dconv, const float* __restrict__ w,
                                                                TA* __restrict__ dqkv, int64_t ld, int chan_off, int B, int heads,
                                                                int T, int H, int W, int Ho, int Wo, int s) {
    constexpr int CW = 16 / sizeof(TA);
    constexpr int NCH = 24 / CW;
    __shared__ __attribute__((aligned(16))) float wsm[27 * 96];
    for (int i = threadIdx.x; i < 27 * 96; i += 256) {
        const int tap = i / 96, c = i - tap * 96;
        wsm[i] = w[c * 27 + tap];
    }
    __syncthreads();
    const int j = threadIdx.x & 3;
    const int64_t Nin = (int64_t)T * H * W;
    const int64_t total = (int64_t)B * heads * Nin;
    const int64_t Lout = (int64_t)T * Ho * Wo;
    for (int64_t it0 = (int64_t)blockIdx.x * 64; it0 < total; it0 += (int64_t)gridDim.x * 64) {
        const int64_t it = it0 + (threadIdx.x >> 2);
        if (it >= total) continue;
        int64_t rem = it;
        const int x = (int)(rem % W); rem /= W;
        const int y = (int)(rem % H); rem /= H;
        const int t = (int)(rem % T); rem /= T;
        const int gh = (int)(rem % heads);
        const int b = (int)(rem / heads);
        const int dy = (y + 1) % s, yo = (y + 1) / s, dx = (x + 1) % s, xo = (x + 1) / s;
        const bool sp_ok = dy < 3 && dx < 3 && yo < Ho && xo < Wo;
        const int yc = sp_ok ? yo : 0, xc = sp_ok ? xo : 0, dyc = sp_ok ? dy : 0, dxc = sp_ok ? dx : 0;
        const TA* pb = dconv + ((int64_t)(b * heads + gh)) * Lout * 96 + ((int64_t)yc * Wo + xc) * 96;
        float4 v[3][NCH * CW / 4];
        float m[3];
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
            const int to = t + 1 - dt;
            const bool ok = sp_ok && to >= 0 && to < T;
            m[dt] = ok ? 1.f : 0.f;
            const TA* p = pb + (int64_t)(ok ? to : 0) * Ho * Wo * 96;
#pragma unroll
            for (int i = 0; i < NCH; ++i)
#pragma unroll
                for (int e = 0; e < CW; e += 4) v[dt][(i * CW + e) / 4] = load4(p + CW * (j + 4 * i) + e);
        }
        float acc[24];
#pragma unroll
        for (int e = 0; e < 24; ++e) acc[e] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
            const float* wt = wsm + ((dt * 3 + dyc) * 3 + dxc) * 96;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c0 = CW * (j + 4 * i);
#pragma unroll
                for (int e = 0; e < CW; e += 4) {
                    const float4 vv = v[dt][(i * CW + e) / 4];
                    float4 ww = *reinterpret_cast<const float4*>(wt + c0 + e);
                    ww.x *= m[dt]; ww.y *= m[dt]; ww.z *= m[dt]; ww.w *= m[dt];
                    acc[i * CW + e] = fmaf(vv.x, ww.x, acc[i * CW + e]);
                    acc[i * CW + e + 1] = fmaf(vv.y, ww.y, acc[i * CW + e + 1]);
                    acc[i * CW + e + 2] = fmaf(vv.z, ww.z, acc[i * CW + e + 2]);
                    acc[i * CW + e + 3] = fmaf(vv.w, ww.w, acc[i * CW + e + 3]);
                }
            }
        }
        TA* o = dqkv + ((int64_t)b * Nin + ((int64_t)t * H + y) * W + x) * ld + chan_off + gh * 96;
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int e = 0; e < CW; e += 4)
                store4(o + CW * (j + 4 * i) + e, make_float4(acc[i * CW + e], acc[i * CW + e + 1], acc[i * CW + e + 2], acc[i * CW + e + 3]));
    }
}

// LayerNorm backward of the pooling conv from what the training forward kept (xhat, rstd): one pass over rows of 96 channels,
// 4 lanes per row.  d_conv = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat)); per-block partial sums of d_gamma = sum dy*xhat
// and d_beta = sum dy go to part[block][192].
template <typename TA>
__global__ __launch_bounds__(256) void pool_ln_bwd_saved_kernel(const TA* __restrict__ xhat, const float* __restrict__ rstd,
                                                                const float* __restrict__ gamma, const TA* __restrict__ dout,
                                                                TA* __restrict__ dconv, float* __restrict__ part, int64_t total,
                                                                const float* __restrict__ gamma2 = nullptr) {
    constexpr int CW = 16 / sizeof(TA);
    constexpr int NCH = 24 / CW;
    __shared__ float red[4][192];        // one row per wave, added in wave order below (LDS float atomics would sum in arrival order)
    if (blockIdx.y == 1) {               // second tensor of a two-tensor launch (k / v pair): `total` tokens further on, own gamma
        xhat += total * 96; dout += total * 96; dconv += total * 96; rstd += total;
        gamma = gamma2;
        part += (int64_t)gridDim.x * 192;
    }
    const int j = threadIdx.x & 3;
    float g[24], dg[24], db[24];
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int e = 0; e < CW; ++e) {
            g[i * CW + e] = gamma[CW * (j + 4 * i) + e];
            dg[i * CW + e] = 0.f;
            db[i * CW + e] = 0.f;
        }
    for (int64_t it0 = (int64_t)blockIdx.x * 64; it0 < total; it0 += (int64_t)gridDim.x * 64) {
        const int64_t it = it0 + (threadIdx.x >> 2);
        const bool ok = it < total;
        const int64_t itc = ok ? it : total - 1;
        const float keep = ok ? 1.f : 0.f;
        float xh[24], dy[24];
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int e = 0; e < CW; e += 4) {
                const float4 a = load4(xhat + itc * 96 + CW * (j + 4 * i) + e);
                const float4 d = load4(dout + itc * 96 + CW * (j + 4 * i) + e);
                const int k = i * CW + e;
                xh[k] = a.x; xh[k + 1] = a.y; xh[k + 2] = a.z; xh[k + 3] = a.w;
                dy[k] = d.x * keep; dy[k + 1] = d.y * keep; dy[k + 2] = d.z * keep; dy[k + 3] = d.w * keep;
            }
        const float r = rstd[itc];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int k = 0; k < 24; ++k) {
            dg[k] = fmaf(dy[k], xh[k], dg[k]);
            db[k] += dy[k];
            dy[k] *= g[k];
            c1 += dy[k];
            c2 = fmaf(dy[k], xh[k], c2);
        }
        c1 += __shfl_xor(c1, 1, 64); c1 += __shfl_xor(c1, 2, 64);
        c2 += __shfl_xor(c2, 1, 64); c2 += __shfl_xor(c2, 2, 64);
        c1 *= (1.0f / 96.0f);
        c2 *= (1.0f / 96.0f);
        if (ok) {
#pragma unroll
            for (int i = 0; i < NCH; ++i)
#pragma unroll
                for (int e = 0; e < CW; e += 4) {
                    const int k = i * CW + e;
                    store4(dconv + it * 96 + CW * (j + 4 * i) + e,
                           make_float4(r * (dy[k] - c1 - xh[k] * c2), r * (dy[k + 1] - c1 - xh[k + 1] * c2),
                                       r * (dy[k + 2] - c1 - xh[k + 2] * c2), r * (dy[k + 3] - c1 - xh[k + 3] * c2)));
                }
        }
    }
    // block reduction: the 64 lanes that share (lane & 3) own the same 24 channels
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int e = 0; e < CW; ++e) {
            const int c = CW * (j + 4 * i) + e;
            float a = dg[i * CW + e], bsum = db[i * CW + e];
#pragma unroll
            for (int off = 4; off < 64; off <<= 1) {       // lanes with the same j inside the wave
                a += __shfl_xor(a, off, 64);
                bsum += __shfl_xor(bsum, off, 64);
            }
            if ((threadIdx.x & 63) < 4) {
                red[threadIdx.x >> 6][c] = a;
                red[threadIdx.x >> 6][96 + c] = bsum;
            }
        }
    __syncthreads();
    if (threadIdx.x < 192)
        part[(int64_t)blockIdx.x * 192 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// dconv: caller-provided scratch, same shape/type as dout.  dqkv slice is fully overwritten.
// dw [96][27] fp32 is ACCUMULATED into (caller zeroes it once per step); dgamma/dbeta: accumulate flag.
// xhat / rstd (optional, from mvit_pool_conv_ln_fwd_train): when given, the LayerNorm backward is one row-wise pass over them
// instead of a second convolution + statistics.
extern "C" int mvit_pool_conv_ln_bwd_saved(const void* qkv, int64_t ld, int chan_off, const float* w, const float* gamma,
                                           const void* xhat, const float* rstd, const void* dout, void* dconv, void* dqkv,
                                           float* dw, float* dgamma, float* dbeta, int accumulate_param, float* workspace, int B,
                                           int heads, int T, int H, int W, int stride_hw, float eps, int act_dtype, void* stream) {
    if ((xhat != nullptr) != (rstd != nullptr)) return MVIT_EINVAL;
    if (!qkv || !w || !gamma || !dout || !dconv || !dqkv || !dw || !dgamma || !dbeta || !workspace || B <= 0 ||
        heads <= 0 || T <= 0 || H <= 0 || W <= 0 || stride_hw <= 0)
        return MVIT_EINVAL;
    if ((ld & 7) || (chan_off & 7)) return MVIT_EUNSUPPORTED;
    if (act_dtype != MVIT_F32 && act_dtype != MVIT_BF16) return MVIT_EDTYPE;
    hipStream_t st = as_stream(stream);
    const int Ho = (H - 1) / stride_hw + 1, Wo = (W - 1) / stride_hw + 1;
    const int64_t tot_out = (int64_t)B * heads * T * Ho * Wo;
    const int64_t tot_in = (int64_t)B * heads * T * H * W;
    int64_t b1 = (tot_out + 63) / 64;
    if (b1 > PB_MAXBLK) b1 = PB_MAXBLK;
    int64_t b2 = (tot_in + 63) / 64;
    if (b2 > 16384) b2 = 16384;
    int64_t b3 = (tot_out + 31) / 32;        // the per-lane token loop is a chain of dependent loads: many short blocks (<= PW_MAXBLK partial rows)
    if (b3 > PW_MAXBLK) b3 = PW_MAXBLK;
    int64_t prow, wrows_;
    pool_bwd_rows(B, heads, H, W, stride_hw, prow, wrows_);
    float* wpart = workspace + prow * 192;     // layout of mvit_pool_bwd_workspace_bytes
    const bool tiled = stride_hw == 1 || stride_hw == 2;
    static const bool dgrad_tiled = getenv("MVIT_POOL_DGRAD_GATHER") == nullptr;
    // the conv weight gradient beside the data gradient on the side stream: -0.55 ms per train step when OFF (profiles/
    // r2_side_stream_ab.txt; it was a gain in round 1, when the weight gradient ended in atomics and a 96-launch reduce tail)
    static const bool pool_bwd_side = getenv("MVIT_POOL_BWD_SIDE") != nullptr && getenv("MVIT_POOL_BWD_SIDE")[0] == '1';
    // d_conv is complete after the first kernel; the conv weight gradient (+ its partial-row reduction) only reads it, so it is
    // issued on the library's side stream and runs beside the d_gamma / d_beta reductions and the data gradient
    // saved statistics + 16-bit build + stride 1 / 2: LayerNorm backward and conv weight gradient in ONE kernel (pool_march.hip); MVIT_POOL_LNB_FUSE=0
    // keeps the row-wise pass + the plain weight-gradient kernel (A/B)
    // (the fused kernel holds 162 registers at stride 1: one workgroup per CU instead of two, which costs what the saved pass returns on the
    // large stride-1 grids -- 107 vs 110 us at 28 x 28, 204 vs 214 at 56 x 56 -- so it takes stride 2, where LDS allows one workgroup per CU
    // either way, and the <= 14 x 14 grids; the rule depends on the geometry alone, never on the batch.  1 = everywhere, 0 = nowhere)
    static const int lnb_mode = getenv("MVIT_POOL_LNB_FUSE") ? atoi(getenv("MVIT_POOL_LNB_FUSE")) : 2;
    const bool lnb_fuse = lnb_mode == 1 || (lnb_mode == 2 && (stride_hw == 2 || (int64_t)Ho * Wo <= 196));
    bool fused_done = false;
    if (xhat && tiled && act_dtype == MVIT_BF16 && lnb_fuse) {
        const int wr = mvit_internal_pool_wgrad_march(qkv, ld, chan_off, dconv, wpart, B, heads, T, H, W, stride_hw, 1, st, xhat, dout, rstd, gamma,
                                                      nullptr, workspace);
        if (wr >= 0) {
            int rr_ = launch_pool_reduce(workspace, wr, 192, dgamma, dbeta, 96, accumulate_param, st, 1);
            if (rr_ != MVIT_OK) return rr_;
            rr_ = launch_pool_reduce(wpart, wr, 2592, dw, dw, 2592, 1, st, 1);
            if (rr_ != MVIT_OK) return rr_;
            fused_done = true;
        } else if (wr != MVIT_EUNSUPPORTED) {
            return wr;
        }
    }
#define RUN(TA)                                                                                                            \
    SideStream* ss = nullptr;                                                                                              \
    hipStream_t sw = st;                                                                                                   \
    if (fused_done) {                                                                                                      \
    } else if (xhat) {                                                                                                     \
        int64_t bs = (tot_out + 63) / 64;       /* as many blocks as the workspace has partial rows for (>= PB_MAXBLK) */  \
        bs = bs > prow ? prow : bs;                                                                                        \
        bs = bs > 2048 ? 2048 : bs;                                                                                        \
        hipLaunchKernelGGL((pool_ln_bwd_saved_kernel<TA>), dim3((unsigned)bs), dim3(256), 0, st, (const TA*)xhat, rstd, gamma, \
                           (const TA*)dout, (TA*)dconv, workspace, tot_out);                                                \
        MVIT_LAUNCH_CHECK();                                                                                               \
        ss = pool_bwd_side ? side_stream_for_current_device() : nullptr;                                                                             \
        if (ss && side_fork(ss, st)) sw = ss->side;                                                                        \
        { const int rr_ = launch_pool_reduce(workspace, (int)bs, 192, dgamma, dbeta, 96, accumulate_param, st, 1); if (rr_ != MVIT_OK) return rr_; } \
    } else if (tiled) {                                                                                                    \
        const int nrows = mvit_internal_pool_ln_bwd_tiled(qkv, ld, chan_off, w, gamma, dout, dconv, workspace, B, heads, T, \
                                                          H, W, stride_hw, eps, act_dtype, st);                            \
        if (nrows < 0) return nrows;                                                                                       \
        ss = pool_bwd_side ? side_stream_for_current_device() : nullptr;                                                                             \
        if (ss && side_fork(ss, st)) sw = ss->side;                                                                        \
        { const int rr_ = launch_pool_reduce(workspace, nrows, 96, dgamma, dgamma, 96, accumulate_param, st); if (rr_ != MVIT_OK) return rr_; } \
        /* d_beta = column sums of dout (workspace rows are free again after the reduce above, same stream) */             \
        const int rcb = mvit_colsum(dout, act_dtype, tot_out, 96, nullptr, 0, dbeta, accumulate_param, workspace, stream);  \
        if (rcb != MVIT_OK) return rcb;                                                                                    \
    } else {                                                                                                               \
        hipLaunchKernelGGL((pool_ln_bwd_kernel<TA>), dim3((unsigned)b1), dim3(256), 0, st, (const TA*)qkv, ld, chan_off, w, \
                           gamma, (const TA*)dout, (TA*)dconv, workspace, B, heads, T, H, W, Ho, Wo, stride_hw, eps);      \
        MVIT_LAUNCH_CHECK();                                                                                               \
        ss = pool_bwd_side ? side_stream_for_current_device() : nullptr;                                                                             \
        if (ss && side_fork(ss, st)) sw = ss->side;                                                                        \
        { const int rr_ = launch_pool_reduce(workspace, (int)b1, 192, dgamma, dbeta, 96, accumulate_param, st); if (rr_ != MVIT_OK) return rr_; } \
    }                                                                                                                      \
    if (fused_done) {                                                                                                      \
    } else if (tiled) {                                                                                                    \
        const int wr = mvit_internal_pool_wgrad_tiled(qkv, ld, chan_off, dconv, wpart, B, heads, T, H, W, stride_hw,        \
                                                      act_dtype, sw);                                                      \
        if (wr < 0) return wr;                                                                                             \
        { const int rr_ = launch_pool_reduce(wpart, wr, 2592, dw, dw, 2592, 1, sw, xhat != nullptr); if (rr_ != MVIT_OK) return rr_; } \
    } else {                                                                                                               \
        hipLaunchKernelGGL((pool_wgrad_kernel<TA>), dim3((unsigned)b3), dim3(1024), 0, sw, (const TA*)qkv, ld, chan_off,     \
                           (const TA*)dconv, wpart, B, heads, T, H, W, Ho, Wo, stride_hw);                                 \
        MVIT_LAUNCH_CHECK();                                                                                               \
        { const int rr_ = launch_pool_reduce(wpart, (int)b3, 2592, dw, dw, 2592, 1, sw, xhat != nullptr); if (rr_ != MVIT_OK) return rr_; } \
    }                                                                                                                      \
    if (stride_hw == 1 && dgrad_tiled) {    /* stride 1: the data gradient IS the tiled convolution with mirrored taps */      \
        const int dr = mvit_internal_pool_dgrad_tiled(dconv, w, dqkv, ld, chan_off, B, heads, T, H, W, act_dtype, st);      \
        if (dr != MVIT_OK) return dr;                                                                                      \
    } else if (stride_hw == 2 && dgrad_tiled) {   /* stride 2: four parity-class convolutions over the d_conv grid, tiled */  \
        const int dr = mvit_internal_pool_dgrad2_tiled(dconv, w, dqkv, ld, chan_off, B, heads, T, H, W, act_dtype, st);     \
        if (dr != MVIT_OK) return dr;                                                                                      \
    } else if (stride_hw >= 3 && dgrad_tiled) {   /* strides >= 3: at most one spatial tap per input token */               \
        hipLaunchKernelGGL((pool_dgrad_sparse_kernel<TA>), dim3((unsigned)b2), dim3(256), 0, st, (const TA*)dconv, w, (TA*)dqkv, \
                           ld, chan_off, B, heads, T, H, W, Ho, Wo, stride_hw);                                            \
        MVIT_LAUNCH_CHECK();                                                                                               \
    } else {                                                                                                               \
        hipLaunchKernelGGL((pool_dgrad_kernel<TA>), dim3((unsigned)b2), dim3(256), 0, st, (const TA*)dconv, w, (TA*)dqkv, ld, \
                           chan_off, B, heads, T, H, W, Ho, Wo, stride_hw);                                                \
        MVIT_LAUNCH_CHECK();                                                                                               \
    }                                                                                                                      \
    if (sw != st && !side_join(ss, st)) return MVIT_ELAUNCH;
    if (act_dtype == MVIT_F32) { RUN(float) } else { RUN(bf16_t) }
#undef RUN
    return MVIT_OK;
}

// Backward of the k and v pooling convs of a block together (stride 2, saved xhat / rstd; other cases: MVIT_EUNSUPPORTED, use the
// single form twice): three launches for both tensors -- LayerNorm backward, conv weight gradient, conv data gradient -- each with
// twice the workgroups of the single form (which at 256 workgroups of 3 waves is latency-bound).  xhat_kv / dout_kv / dconv_kv are
// [2][B][heads][T*Ho*Wo][96] (k then v), rstd_kv [2][...]; workspace >= 2 x mvit_pool_bwd_workspace_bytes(...).  The per-tensor
// sums (d_w, d_gamma, d_beta) are reduced over each tensor's own contiguous partial rows: bit-identical to the single form.
extern "C" int mvit_pool_conv_ln_bwd_saved_kv(const void* qkv, int64_t ld, int chan_off_k, const float* w_k, const float* gamma_k,
                                              const float* w_v, const float* gamma_v, const void* xhat_kv, const float* rstd_kv,
                                              const void* dout_kv, void* dconv_kv, void* dqkv, float* dw_k, float* dgamma_k, float* dbeta_k,
                                              float* dw_v, float* dgamma_v, float* dbeta_v, int accumulate_param, float* workspace, int B,
                                              int heads, int T, int H, int W, int stride_hw, int act_dtype, void* stream) {
    if (!qkv || !w_k || !gamma_k || !w_v || !gamma_v || !xhat_kv || !rstd_kv || !dout_kv || !dconv_kv || !dqkv || !dw_k || !dgamma_k ||
        !dbeta_k || !dw_v || !dgamma_v || !dbeta_v || !workspace || B <= 0 || heads <= 0 || T <= 0 || H <= 0 || W <= 0)
        return MVIT_EINVAL;
    if (act_dtype != MVIT_F32 && act_dtype != MVIT_BF16) return MVIT_EDTYPE;
    if ((ld & 7) || (chan_off_k & 7)) return MVIT_EUNSUPPORTED;
    if (stride_hw != 2 || (int64_t)2 * B * heads > 65535) return MVIT_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int64_t tot_out = (int64_t)B * heads * T * Ho * Wo;            // tokens of ONE tensor
    int64_t prow, wrows_;                   // rows of mvit_pool_bwd_workspace_bytes, per tensor
    pool_bwd_rows(B, heads, H, W, 2, prow, wrows_);
    int64_t bs = (tot_out + 63) / 64;
    bs = bs > prow ? prow : bs;
    bs = bs > 2048 ? 2048 : bs;
    float* wpart = workspace + 2 * prow * 192;
    static const bool lnb_fuse_kv = !(getenv("MVIT_POOL_LNB_FUSE") && getenv("MVIT_POOL_LNB_FUSE")[0] == '0');
    if (act_dtype == MVIT_BF16 && lnb_fuse_kv) {      // LayerNorm backward + weight gradient of both tensors in one launch (pool_march.hip)
        const int wr = mvit_internal_pool_wgrad_march(qkv, ld, chan_off_k, dconv_kv, wpart, B, heads, T, H, W, 2, 2, st, xhat_kv, dout_kv, rstd_kv,
                                                      gamma_k, gamma_v, workspace);
        if (wr >= 0) {            // rows are set-major: the first wr / 2 belong to k
            int rc2 = launch_pool_reduce(workspace, wr / 2, 192, dgamma_k, dbeta_k, 96, accumulate_param, st, 1);
            if (rc2 != MVIT_OK) return rc2;
            rc2 = launch_pool_reduce(workspace + (int64_t)(wr / 2) * 192, wr / 2, 192, dgamma_v, dbeta_v, 96, accumulate_param, st, 1);
            if (rc2 != MVIT_OK) return rc2;
            rc2 = launch_pool_reduce(wpart, wr / 2, 2592, dw_k, dw_k, 2592, 1, st, 1);
            if (rc2 != MVIT_OK) return rc2;
            rc2 = launch_pool_reduce(wpart + (int64_t)(wr / 2) * 2592, wr / 2, 2592, dw_v, dw_v, 2592, 1, st, 1);
            if (rc2 != MVIT_OK) return rc2;
            return mvit_internal_pool_dgrad2_tiled_kv(dconv_kv, w_k, w_v, dqkv, ld, chan_off_k, B, heads, T, H, W, act_dtype, st);
        }
        if (wr != MVIT_EUNSUPPORTED) return wr;
    }
#define RUNKV(TA)                                                                                                                  \
    hipLaunchKernelGGL((pool_ln_bwd_saved_kernel<TA>), dim3((unsigned)bs, 2), dim3(256), 0, st, (const TA*)xhat_kv, rstd_kv, gamma_k, \
                       (const TA*)dout_kv, (TA*)dconv_kv, workspace, tot_out, gamma_v);                                            \
    MVIT_LAUNCH_CHECK();
    if (act_dtype == MVIT_F32) { RUNKV(float) } else { RUNKV(bf16_t) }
#undef RUNKV
    int rc = launch_pool_reduce(workspace, (int)bs, 192, dgamma_k, dbeta_k, 96, accumulate_param, st, 1);
    if (rc != MVIT_OK) return rc;
    rc = launch_pool_reduce(workspace + bs * 192, (int)bs, 192, dgamma_v, dbeta_v, 96, accumulate_param, st, 1);
    if (rc != MVIT_OK) return rc;
    const int wr = mvit_internal_pool_wgrad_tiled_kv(qkv, ld, chan_off_k, dconv_kv, wpart, B, heads, T, H, W, act_dtype, st);
    if (wr < 0) return wr;
    rc = launch_pool_reduce(wpart, wr / 2, 2592, dw_k, dw_k, 2592, 1, st, 1);
    if (rc != MVIT_OK) return rc;
    rc = launch_pool_reduce(wpart + (int64_t)(wr / 2) * 2592, wr / 2, 2592, dw_v, dw_v, 2592, 1, st, 1);
    if (rc != MVIT_OK) return rc;
    return mvit_internal_pool_dgrad2_tiled_kv(dconv_kv, w_k, w_v, dqkv, ld, chan_off_k, B, heads, T, H, W, act_dtype, st);
}

extern "C" int mvit_pool_conv_ln_bwd(const void* qkv, int64_t ld, int chan_off, const float* w, const float* gamma,
                                     const void* dout, void* dconv, void* dqkv, float* dw, float* dgamma, float* dbeta,
                                     int accumulate_param, float* workspace, int B, int heads, int T, int H, int W,
                                     int stride_hw, float eps, int act_dtype, void* stream) {
    return mvit_pool_conv_ln_bwd_saved(qkv, ld, chan_off, w, gamma, nullptr, nullptr, dout, dconv, dqkv, dw, dgamma, dbeta,
                                       accumulate_param, workspace, B, heads, T, H, W, stride_hw, eps, act_dtype, stream);
}
