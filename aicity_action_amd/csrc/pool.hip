// Pooling conv (depthwise 3x3x3, stride (1,s,s), zero pad 1, weight shared across heads) fused with the
// LayerNorm(96, eps) that follows it; reads the fused qkv activation in place (no head-split copies).
// HBM/L2-bound stencil: 4 lanes per output token, 24 channels per lane as 16-byte chunks
// (chunk index = lane + 4*i, so one wave-instruction reads 64 contiguous bytes per token).
#include "common.h"

template <typename TA>
__global__ __launch_bounds__(256) void pool_conv_ln_kernel(const TA* __restrict__ qkv, int64_t ld, int chan_off,
                                                           const float* __restrict__ w, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, TA* __restrict__ out,
                                                           TA* __restrict__ xhat, float* __restrict__ rstd_out, int B,
                                                           int heads, int T, int H, int W, int Ho, int Wo, int s,
                                                           float eps) {
    constexpr int CW = 16 / sizeof(TA);  // channels per 16-byte chunk (bf16: 8, fp32: 4)
    constexpr int NCH = 24 / CW;         // chunks per lane
    __shared__ __attribute__((aligned(16))) float wsm[27 * 96];  // [tap][channel]
    for (int i = threadIdx.x; i < 27 * 96; i += 256) {
        const int tap = i / 96, c = i - tap * 96;
        wsm[i] = w[c * 27 + tap];
    }
    __syncthreads();
    const int j = threadIdx.x & 3;
    float g[24], bt[24];
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int e = 0; e < CW; ++e) {
            g[i * CW + e] = gamma[CW * (j + 4 * i) + e];
            bt[i * CW + e] = beta[CW * (j + 4 * i) + e];
        }
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
                // the three dx taps of this row: loads issued together at clamped x (taps outside the frame are zeroed after
                // the load) -- with `continue` on the x test every 16-byte load sat behind its own branch and vmcnt(0)
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
                            const float4 w0 = *reinterpret_cast<const float4*>(wt + c0 + e);
                            acc[i * CW + e + 0] = fmaf(xv[dx][i * CW + e + 0] * msk[dx], w0.x, acc[i * CW + e + 0]);
                            acc[i * CW + e + 1] = fmaf(xv[dx][i * CW + e + 1] * msk[dx], w0.y, acc[i * CW + e + 1]);
                            acc[i * CW + e + 2] = fmaf(xv[dx][i * CW + e + 2] * msk[dx], w0.z, acc[i * CW + e + 2]);
                            acc[i * CW + e + 3] = fmaf(xv[dx][i * CW + e + 3] * msk[dx], w0.w, acc[i * CW + e + 3]);
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
        for (int e = 0; e < 24; ++e) {
            acc[e] -= mean;
            sq += acc[e] * acc[e];
        }
        sq += __shfl_xor(sq, 1, 64);
        sq += __shfl_xor(sq, 2, 64);
        const float rstd = 1.0f / sqrtf(sq * (1.0f / 96.0f) + eps);
        if (ok) {
            TA* o = out + it * 96;
            if (xhat) {
#pragma unroll
                for (int i = 0; i < NCH; ++i)
#pragma unroll
                    for (int q4 = 0; q4 < CW / 4; ++q4) {
                        const int e = i * CW + q4 * 4;
                        store4(xhat + it * 96 + CW * (j + 4 * i) + q4 * 4, make_float4(acc[e] * rstd, acc[e + 1] * rstd, acc[e + 2] * rstd, acc[e + 3] * rstd));
                    }
                if (j == 0) rstd_out[it] = rstd;
            }
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c0 = CW * (j + 4 * i);
#pragma unroll
                for (int q4 = 0; q4 < CW / 4; ++q4) {
                    float4 v;
                    const int e = i * CW + q4 * 4;
                    v.x = acc[e + 0] * rstd * g[e + 0] + bt[e + 0];
                    v.y = acc[e + 1] * rstd * g[e + 1] + bt[e + 1];
                    v.z = acc[e + 2] * rstd * g[e + 2] + bt[e + 2];
                    v.w = acc[e + 3] * rstd * g[e + 3] + bt[e + 3];
                    store4(o + c0 + q4 * 4, v);
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Tiled variant for spatial stride 1 and 2 (98 % of the pooled tokens).
//
// One workgroup owns an 8-wide x ROWS-high output tile of one (batch, head) and marches over the T
// input frames: the current input frame's halo tile lives in LDS (double-buffered, next frame
// prefetched through registers), every thread owns one channel PAIR (its 27x2 weights stay in
// registers) and one output row of 8 tokens, and keeps three rolling accumulator sets (output frames
// f-1, f, f+1) so each LDS value is read once per (dy) and used for 3 dt x 3 dx taps.  A finished frame
// is staged in LDS as fp32, LayerNorm'ed by 4 lanes per token and stored as whole token rows.
// HBM-bound target; VALU (fp32 FMA) is the co-limit: 2592 FMA per output token.
// ------------------------------------------------------------------------------------------------
template <typename TA>
__device__ __forceinline__ void load2(const TA* p, float& a, float& b);
template <>
__device__ __forceinline__ void load2<bf16_t>(const bf16_t* p, float& a, float& b) {
    const uint32_t u = *reinterpret_cast<const uint32_t*>(p);
    a = lo16_to_f32(u);
    b = hi16_to_f32(u);
}
template <>
__device__ __forceinline__ void load2<float>(const float* p, float& a, float& b) {
    const float2 u = *reinterpret_cast<const float2*>(p);
    a = u.x;
    b = u.y;
}

template <typename TA>
__device__ __forceinline__ void store2(TA* p, float a, float b);
template <>
__device__ __forceinline__ void store2<bf16_t>(bf16_t* p, float a, float b) { *reinterpret_cast<uint32_t*>(p) = pack_bf16x2(a, b); }
template <>
__device__ __forceinline__ void store2<float>(float* p, float a, float b) { *reinterpret_cast<float2*>(p) = make_float2(a, b); }

#ifndef POOL_ABL
#define POOL_ABL 0      // timing ablations (tools): 1 no conv arithmetic, 2 no LayerNorm finalize, 4 no tile loads
#endif
template <typename TA, int S>
struct PoolTile {
    static constexpr int ROWS = (S == 1) ? 8 : 4;
    static constexpr int XO = 8;
    static constexpr int IH = S * (ROWS - 1) + 3;
    static constexpr int IW = S * (XO - 1) + 3;
    static constexpr int NT = 48 * ROWS;
    static constexpr int NTOK = ROWS * XO;
    static constexpr int CW = 16 / sizeof(TA);
    static constexpr int CPT = 96 / CW;                 // 16-byte chunks per token
    static constexpr int NCHUNK = IH * IW * CPT;
    static constexpr int PF = (NCHUNK + NT - 1) / NT;   // prefetch registers (uint4) per thread
    static constexpr int IN_BYTES = IH * IW * 96 * (int)sizeof(TA);
    static constexpr int W_BYTES = 27 * 96 * 4;          // weights [tap][channel] fp32
    static constexpr bool DB = false;                   // single input tile + register prefetch: 54 KB -> 3 workgroups per CU
    static constexpr int NBUF = DB ? 2 : 1;
    static constexpr int SMEM = NBUF * IN_BYTES + NTOK * 96 * 4 + W_BYTES;
};

// BWD = true: the same march recomputes the conv + LayerNorm statistics and applies the LayerNorm BACKWARD in the finalize
// step: `out` then receives d_conv (gradient wrt the conv output), `dout` is the incoming gradient, and per-block partial
// sums of d_gamma / d_beta go to part[block][192] (accumulated in LDS with ds_add_f32).
// PLAIN = true (stride 1): the bare convolution with the taps mirrored and no LayerNorm, written token-major into a slice of a
// [B][tokens][out_ld] buffer -- the DATA gradient of the stride-1 pooling conv (input d_conv [B*heads][tokens][96] passed as
// `qkv` with ld = 96, heads = 1; out_heads = the real head count for the output slice).
template <typename TA, int S, bool BWD, bool PLAIN = false>
__global__ __launch_bounds__(S == 1 ? 384 : 192, 3) void pool_tiled_kernel(
    const TA* __restrict__ qkv, int64_t ld, int chan_off, const float* __restrict__ w, const float* __restrict__ gamma,
    const float* __restrict__ beta, TA* __restrict__ out, const TA* __restrict__ dout, float* __restrict__ part, int heads,
    int T, int H, int W, int Ho, int Wo, float eps, int64_t out_ld = 0, int out_chan_off = 0, int out_heads = 1,
    int set_bh = 0, const float* __restrict__ w2 = nullptr, const float* __restrict__ gamma2 = nullptr,
    const float* __restrict__ beta2 = nullptr) {
    // set_bh > 0: TWO tensors in one launch (the k and the v pooling conv of a block: adjacent head groups of the fused qkv buffer,
    // own conv weights / LayerNorm parameters, outputs back to back): blockIdx.y = set * set_bh + (b * heads + g).  One such launch
    // of the steady 384-wide blocks is 512 workgroups instead of two latency-bound launches of 256 (38 us instead of 2 x 30).
    using P = PoolTile<TA, S>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float dgam = 0.f;   // BWD: threads < 96 own one channel of the d_gamma partial (d_beta = column sum of dout, done by the caller)
    float* stage = reinterpret_cast<float*>(smem + P::NBUF * P::IN_BYTES);

    const int tid = threadIdx.x;
    const int cp = tid % 48, row = tid / 48;
    const int tiles_x = (Wo + P::XO - 1) / P::XO;
    const int tx0 = (blockIdx.x % tiles_x) * P::XO, ty0 = (blockIdx.x / tiles_x) * P::ROWS;
    const int bh = blockIdx.y;
    int bhs = bh, hoff = 0;
    if (set_bh > 0 && bh >= set_bh) { bhs = bh - set_bh; hoff = heads; w = w2; gamma = gamma2; beta = beta2; }
    const int b = bhs / heads, g = bhs - b * heads;
    const int64_t Nin = (int64_t)T * H * W;
    const TA* base = qkv + (int64_t)b * Nin * ld + chan_off + (hoff + g) * 96;
    const int y_in0 = S * ty0 - 1, x_in0 = S * tx0 - 1;

    float* wl = reinterpret_cast<float*>(smem + P::NBUF * P::IN_BYTES + P::NTOK * 96 * 4);
    for (int i = tid; i < 27 * 96; i += P::NT) {
        const int tap = i / 96, c = i - tap * 96;
        wl[i] = w[c * 27 + (PLAIN ? 26 - tap : tap)];
    }
    const float* wmine = wl + 2 * cp;
    // LayerNorm lane mapping (threads < NTOK*4): token = tid>>2, j = tid&3, 24 channels in CW-wide chunks j+4i
    const int lj = tid & 3, ltok = tid >> 2;
    constexpr int CW = P::CW, NCH = 24 / CW;

    uint4 pf[P::PF];
    int poff[P::PF];   // element offset of this thread's chunk inside one frame, -1 = zero padding / unused
#pragma unroll
    for (int i = 0; i < P::PF; ++i) {
        const int c = tid + P::NT * i;
        poff[i] = -1;
        if (c < P::NCHUNK) {
            const int tok = c / P::CPT, ch = c - tok * P::CPT;
            const int iy = tok / P::IW, ix = tok - iy * P::IW;
            const int y = y_in0 + iy, x = x_in0 + ix;
            if (y >= 0 && y < H && x >= 0 && x < W) poff[i] = (int)((y * W + x) * ld) + ch * CW;
        }
    }
    const int frame_stride = (int)((int64_t)H * W * ld);
    auto prefetch = [&](int f) {
        const TA* fb = base + (int64_t)f * frame_stride;
#pragma unroll
        for (int i = 0; i < P::PF; ++i)       // unconditional (clamped) loads, all in flight together; padding is zeroed at commit
            pf[i] = *reinterpret_cast<const uint4*>(fb + (poff[i] >= 0 ? poff[i] : 0));
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < P::PF; ++i) {
            const int c = tid + P::NT * i;
            if (c < P::NCHUNK) *reinterpret_cast<uint4*>(smem + buf * P::IN_BYTES + c * 16) = poff[i] >= 0 ? pf[i] : make_uint4(0, 0, 0, 0);
        }
    };

    float acc[3][P::XO][2];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int x = 0; x < P::XO; ++x) acc[a][x][0] = acc[a][x][1] = 0.f;

    auto finalize = [&](int fo) {   // acc[0] holds output frame fo; all threads call this (barriers inside)
#pragma unroll
        for (int x = 0; x < P::XO; ++x)
            *reinterpret_cast<float2*>(stage + (row * P::XO + x) * 96 + 2 * cp) = make_float2(acc[0][x][0], acc[0][x][1]);
        __syncthreads();
        if (tid < P::NTOK * 4) {
            float v[24];
#pragma unroll
            for (int i = 0; i < NCH; ++i)
#pragma unroll
                for (int e = 0; e < CW; e += 4) {
                    const float4 t4 = *reinterpret_cast<const float4*>(stage + ltok * 96 + CW * (lj + 4 * i) + e);
                    v[i * CW + e] = t4.x; v[i * CW + e + 1] = t4.y; v[i * CW + e + 2] = t4.z; v[i * CW + e + 3] = t4.w;
                }
            if constexpr (PLAIN) {
                const int yo = ty0 + ltok / P::XO, xo = tx0 + ltok % P::XO;
                if (yo < Ho && xo < Wo) {
                    const int ob = bh / out_heads, og = bh - ob * out_heads;
                    TA* o = out + ((int64_t)ob * T * Ho * Wo + ((int64_t)fo * Ho + yo) * Wo + xo) * out_ld + out_chan_off + og * 96;
#pragma unroll
                    for (int i = 0; i < NCH; ++i)
#pragma unroll
                        for (int e = 0; e < CW; e += 4)
                            store4(o + CW * (lj + 4 * i) + e, make_float4(v[i * CW + e], v[i * CW + e + 1], v[i * CW + e + 2], v[i * CW + e + 3]));
                }
                return;
            }
            float sum = 0.f;
#pragma unroll
            for (int e = 0; e < 24; ++e) sum += v[e];
            sum += __shfl_xor(sum, 1, 64);
            sum += __shfl_xor(sum, 2, 64);
            const float mean = sum * (1.0f / 96.0f);
            float sq = 0.f;
#pragma unroll
            for (int e = 0; e < 24; ++e) {
                v[e] -= mean;
                sq += v[e] * v[e];
            }
            sq += __shfl_xor(sq, 1, 64);
            sq += __shfl_xor(sq, 2, 64);
            const float rstd = 1.0f / sqrtf(sq * (1.0f / 96.0f) + eps);
            const int yo = ty0 + ltok / P::XO, xo = tx0 + ltok % P::XO;
            const bool tok_ok = yo < Ho && xo < Wo;
            const int64_t orow = (((int64_t)bh * T + fo) * Ho * Wo + (int64_t)yo * Wo + xo) * 96;
            if (BWD) {
                float dyv[24];
                // unconditional loads at a clamped row (a bounds test per load makes the compiler wait after each one), masked after
                const int64_t lrow = tok_ok ? orow : 0;
                const float keep = tok_ok ? 1.f : 0.f;
                if constexpr (S != 1) {                // stride-2 tiles: the batched form spills (measured slower); load piece by piece
#pragma unroll
                    for (int i = 0; i < NCH; ++i)
#pragma unroll
                        for (int e = 0; e < CW; e += 4) {
                            float4 d4 = make_float4(0.f, 0.f, 0.f, 0.f);
                            if (tok_ok) d4 = load4(dout + orow + CW * (lj + 4 * i) + e);
                            dyv[i * CW + e] = d4.x; dyv[i * CW + e + 1] = d4.y; dyv[i * CW + e + 2] = d4.z; dyv[i * CW + e + 3] = d4.w;
                        }
                } else if constexpr (sizeof(TA) == 2) {       // 16-bit: keep the six 8-byte pieces packed until all are requested
                    uint2 raw[NCH * CW / 4];
#pragma unroll
                    for (int i = 0; i < NCH; ++i)
#pragma unroll
                        for (int e = 0; e < CW; e += 4) raw[(i * CW + e) / 4] = *reinterpret_cast<const uint2*>(dout + lrow + CW * (lj + 4 * i) + e);
#pragma unroll
                    for (int k4 = 0; k4 < NCH * CW / 4; ++k4) {
                        dyv[4 * k4] = lo16_to_f32(raw[k4].x) * keep; dyv[4 * k4 + 1] = hi16_to_f32(raw[k4].x) * keep;
                        dyv[4 * k4 + 2] = lo16_to_f32(raw[k4].y) * keep; dyv[4 * k4 + 3] = hi16_to_f32(raw[k4].y) * keep;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < NCH; ++i)
#pragma unroll
                        for (int e = 0; e < CW; e += 4) {
                            const float4 d4 = load4(dout + lrow + CW * (lj + 4 * i) + e);
                            dyv[i * CW + e] = d4.x * keep; dyv[i * CW + e + 1] = d4.y * keep; dyv[i * CW + e + 2] = d4.z * keep; dyv[i * CW + e + 3] = d4.w * keep;
                        }
                }
                float c1 = 0.f, c2 = 0.f;
#pragma unroll
                for (int i = 0; i < NCH; ++i)
#pragma unroll
                    for (int e = 0; e < CW; ++e) {
                        const int k = i * CW + e, c = CW * (lj + 4 * i) + e;
                        v[k] *= rstd;                                   // xhat
                        stage[ltok * 96 + c] = dyv[k] * v[k];           // dy * xhat -> column sums below (dy = 0 for padded tokens)
                        dyv[k] *= gamma[c];
                        c1 += dyv[k];
                        c2 += dyv[k] * v[k];
                    }
                c1 += __shfl_xor(c1, 1, 64); c1 += __shfl_xor(c1, 2, 64);
                c2 += __shfl_xor(c2, 1, 64); c2 += __shfl_xor(c2, 2, 64);
                c1 *= (1.0f / 96.0f);
                c2 *= (1.0f / 96.0f);
                if (tok_ok) {
#pragma unroll
                    for (int i = 0; i < NCH; ++i)
#pragma unroll
                        for (int e = 0; e < CW; e += 4) {
                            const int k = i * CW + e;
                            float4 r;
                            r.x = rstd * (dyv[k] - c1 - v[k] * c2);
                            r.y = rstd * (dyv[k + 1] - c1 - v[k + 1] * c2);
                            r.z = rstd * (dyv[k + 2] - c1 - v[k + 2] * c2);
                            r.w = rstd * (dyv[k + 3] - c1 - v[k + 3] * c2);
                            store4(out + orow + CW * (lj + 4 * i) + e, r);
                        }
                }
            } else if (tok_ok) {
                TA* o = out + orow;
                if (dout) {     // training forward: keep xhat (and rstd) so that the backward needs no second convolution
                    TA* xh = const_cast<TA*>(dout) + orow;
#pragma unroll
                    for (int i = 0; i < NCH; ++i)
#pragma unroll
                        for (int e = 0; e < CW; e += 4)
                            store4(xh + CW * (lj + 4 * i) + e, make_float4(v[i * CW + e] * rstd, v[i * CW + e + 1] * rstd, v[i * CW + e + 2] * rstd, v[i * CW + e + 3] * rstd));
                    if (lj == 0) part[orow / 96] = rstd;
                }
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    const int c0 = CW * (lj + 4 * i);
#pragma unroll
                    for (int e = 0; e < CW; e += 4) {
                        float4 r;
                        r.x = v[i * CW + e] * rstd * gamma[c0 + e] + beta[c0 + e];
                        r.y = v[i * CW + e + 1] * rstd * gamma[c0 + e + 1] + beta[c0 + e + 1];
                        r.z = v[i * CW + e + 2] * rstd * gamma[c0 + e + 2] + beta[c0 + e + 2];
                        r.w = v[i * CW + e + 3] * rstd * gamma[c0 + e + 3] + beta[c0 + e + 3];
                        store4(o + c0 + e, r);
                    }
                }
            }
        }
        if (BWD) {
            __syncthreads();
            if (tid < 96) {
#pragma unroll 8
                for (int tk = 0; tk < P::NTOK; ++tk) dgam += stage[tk * 96 + tid];
            }
        }
    };

    prefetch(0);
    commit(0);
    __syncthreads();
    for (int f = 0; f < T; ++f) {
        if (P::DB && f + 1 < T) prefetch(f + 1);
        const int cur = P::DB ? (f & 1) : 0;
        const TA* tile = reinterpret_cast<const TA*>(smem + cur * P::IN_BYTES) + 2 * cp;
        const lds_cptr_t wm = lds_opaque(wmine);   // keep the 54 weight reads inside the loop (LICM would pin 54 VGPRs), as LDS reads
#pragma unroll 1
        for (int dy = 0; dy < ((POOL_ABL & 1) ? 0 : 3); ++dy) {
            float xin[P::IW][2];
            const TA* rp = tile + (S * row + dy) * P::IW * 96;
#pragma unroll
            for (int ix = 0; ix < P::IW; ++ix) load2<TA>(rp + ix * 96, xin[ix][0], xin[ix][1]);
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {
                // input frame f is tap dt of output frame f + 1 - dt  -> accumulator set 2 - dt
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const lds_f32x2_t wv = lds_ld<lds_f32x2_t>(wm, ((dt * 3 + dy) * 3 + dx) * 96 * 4);
                    const float w0 = wv.x, w1 = wv.y;
#pragma unroll
                    for (int x = 0; x < P::XO; ++x) {
                        acc[2 - dt][x][0] = fmaf(w0, xin[S * x + dx][0], acc[2 - dt][x][0]);
                        acc[2 - dt][x][1] = fmaf(w1, xin[S * x + dx][1], acc[2 - dt][x][1]);
                    }
                }
            }
        }
        // acc[0] = output frame f-1 is complete
        if (f >= 1 && !(POOL_ABL & 2)) finalize(f - 1);
        else __syncthreads();
        if (f + 1 < T) {
            // (the loads are issued here, not ahead of this frame's arithmetic: held across it the ten prefetch registers push the kernel
            // over its 168-register budget -- spills -- and measured 7-11 % slower, r3; the co-resident workgroups hide the round trip)
            if (!P::DB && !(POOL_ABL & 4)) prefetch(f + 1);   // single buffer: every thread is past its tile reads (barrier above)
            commit(P::DB ? ((f + 1) & 1) : 0);
        }
        __syncthreads();
#pragma unroll
        for (int x = 0; x < P::XO; ++x) {
            acc[0][x][0] = acc[1][x][0]; acc[0][x][1] = acc[1][x][1];
            acc[1][x][0] = acc[2][x][0]; acc[1][x][1] = acc[2][x][1];
            acc[2][x][0] = 0.f; acc[2][x][1] = 0.f;
        }
    }
    finalize(T - 1);
    if (BWD && tid < 96) part[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 96 + tid] = dgam;
}

template <typename TA, int S>
static int launch_pool_tiled(const void* qkv, int64_t ld, int chan_off, const float* w, const float* gamma,
                             const float* beta, void* out, void* xhat, float* rstd, int B, int heads, int T, int H, int W, int Ho, int Wo,
                             float eps, hipStream_t st, const float* w2 = nullptr, const float* gamma2 = nullptr,
                             const float* beta2 = nullptr) {
    using P = PoolTile<TA, S>;
    const int nset = w2 ? 2 : 1;
    dim3 grid(((Wo + P::XO - 1) / P::XO) * ((Ho + P::ROWS - 1) / P::ROWS), nset * B * heads);
    static DevFlags attr_done_tab; DevFlag attr_done = dev_flag(attr_done_tab);
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pool_tiled_kernel<TA, S, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, P::SMEM) != hipSuccess)
            return MVIT_ELAUNCH;
        attr_done = true;
    }
    hipLaunchKernelGGL((pool_tiled_kernel<TA, S, false>), grid, dim3(P::NT), P::SMEM, st, (const TA*)qkv, ld, chan_off, w, gamma,
                       beta, (TA*)out, (const TA*)xhat, rstd, heads, T, H, W, Ho, Wo, eps, (int64_t)0, 0, 1, nset == 2 ? B * heads : 0, w2,
                       gamma2, beta2);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}

template <typename TA, int S>
static int launch_pool_tiled_bwd(const void* qkv, int64_t ld, int chan_off, const float* w, const float* gamma, const void* dout,
                                 void* dconv, float* part, int B, int heads, int T, int H, int W, int Ho, int Wo, float eps,
                                 hipStream_t st) {
    using P = PoolTile<TA, S>;
    dim3 grid(((Wo + P::XO - 1) / P::XO) * ((Ho + P::ROWS - 1) / P::ROWS), B * heads);
    static DevFlags attr_done_tab; DevFlag attr_done = dev_flag(attr_done_tab);
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pool_tiled_kernel<TA, S, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, P::SMEM) != hipSuccess)
            return MVIT_ELAUNCH;
        attr_done = true;
    }
    hipLaunchKernelGGL((pool_tiled_kernel<TA, S, true>), grid, dim3(P::NT), P::SMEM, st, (const TA*)qkv, ld, chan_off, w, gamma,
                       gamma, (TA*)dconv, (const TA*)dout, part, heads, T, H, W, Ho, Wo, eps);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the pooling conv through the same tiled march:
//   dw[c][dt][dy][dx] = sum_tokens d_conv[(to,yo,xo)][c] * in[(to+dt-1, S*yo+dy-1, S*xo+dx-1)][c]
// LDS holds the current input frame's halo tile and a 3-deep ring of d_conv frame tiles (f-1, f, f+1); thread =
// (channel pair, output row) keeps its 27x2 partial sums in registers over all frames; combined per block with
// ds_add_f32, one partial row [2592] per workgroup, reduced by pool_reduce in pool_bwd.hip.
// ------------------------------------------------------------------------------------------------
template <typename TA, int S>
__global__ __launch_bounds__(S == 1 ? 384 : 192, 2) void pool_wgrad_tiled_kernel(
    const TA* __restrict__ qkv, int64_t ld, int chan_off, const TA* __restrict__ dconv, float* __restrict__ part, int heads,
    int T, int H, int W, int Ho, int Wo, int set_bh = 0) {
    using P = PoolTile<TA, S>;
    constexpr int DT_BYTES = P::NTOK * 96 * (int)sizeof(TA);     // one d_conv frame tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* in_lds = smem;                                         // [IH*IW][96]
    char* dc_lds = smem + P::IN_BYTES;                           // [3][NTOK][96]
    const int tid = threadIdx.x;
    const int cp = tid % 48, row = tid / 48;
    const int tiles_x = (Wo + P::XO - 1) / P::XO;
    const int tx0 = (blockIdx.x % tiles_x) * P::XO, ty0 = (blockIdx.x / tiles_x) * P::ROWS;
    const int bh = blockIdx.y;                  // set_bh > 0: two tensors (see pool_tiled_kernel), set-major -> each set's partial rows are contiguous
    const int hoff = (set_bh > 0 && bh >= set_bh) ? heads : 0, bhs = hoff ? bh - set_bh : bh;
    const int b = bhs / heads, g = bhs - b * heads;
    const int64_t Nin = (int64_t)T * H * W;
    const TA* base = qkv + (int64_t)b * Nin * ld + chan_off + (hoff + g) * 96;
    const TA* dbase = dconv + (int64_t)bh * T * Ho * Wo * 96;
    const int y_in0 = S * ty0 - 1, x_in0 = S * tx0 - 1;
    constexpr int CW = P::CW;

    // The next frame's input tile and d_conv tile are requested into registers BEFORE the current frame's arithmetic (unconditional
    // loads at clamped positions, padding zeroed when they are written to LDS after it): with bounds-tested loads between two
    // barriers every frame paid 5 dependent round trips to memory.
    constexpr int PFI = P::PF, PFD = (P::NTOK * P::CPT + P::NT - 1) / P::NT;
    uint4 pin[PFI], pdc[PFD];
    int ioff[PFI], doff[PFD];          // element offsets inside one frame, -1 = padding / unused
#pragma unroll
    for (int i = 0; i < PFI; ++i) {
        const int c = tid + P::NT * i;
        ioff[i] = -1;
        if (c < P::NCHUNK) {
            const int tok = c / P::CPT, ch = c - tok * P::CPT;
            const int iy = tok / P::IW, ix = tok - iy * P::IW;
            const int y = y_in0 + iy, x = x_in0 + ix;
            if (y >= 0 && y < H && x >= 0 && x < W) ioff[i] = (int)((y * W + x) * ld) + ch * CW;
        }
    }
#pragma unroll
    for (int i = 0; i < PFD; ++i) {
        const int c = tid + P::NT * i;
        doff[i] = -1;
        if (c < P::NTOK * P::CPT) {
            const int tok = c / P::CPT, ch = c - tok * P::CPT;
            const int yo = ty0 + tok / P::XO, xo = tx0 + tok % P::XO;
            if (yo < Ho && xo < Wo) doff[i] = (yo * Wo + xo) * 96 + ch * CW;
        }
    }
    const int64_t in_frame = (int64_t)H * W * ld, dc_frame = (int64_t)Ho * Wo * 96;
    auto pre_in = [&](int f) {
        const TA* fb = base + f * in_frame;
#pragma unroll
        for (int i = 0; i < PFI; ++i) pin[i] = *reinterpret_cast<const uint4*>(fb + (ioff[i] >= 0 ? ioff[i] : 0));
    };
    auto commit_in = [&]() {
#pragma unroll
        for (int i = 0; i < PFI; ++i) {
            const int c = tid + P::NT * i;
            if (c < P::NCHUNK) *reinterpret_cast<uint4*>(in_lds + c * 16) = ioff[i] >= 0 ? pin[i] : make_uint4(0, 0, 0, 0);
        }
    };
    auto pre_dc = [&](int fo) {      // frames outside [0, T) read frame 0 and are zeroed at commit
        const TA* fb = dbase + (fo >= 0 && fo < T ? fo : 0) * dc_frame;
#pragma unroll
        for (int i = 0; i < PFD; ++i) pdc[i] = *reinterpret_cast<const uint4*>(fb + (doff[i] >= 0 ? doff[i] : 0));
    };
    auto commit_dc = [&](int fo) {   // d_conv frame fo -> ring slot fo % 3
        char* dst = dc_lds + ((fo + 3) % 3) * DT_BYTES;
        const bool fok = fo >= 0 && fo < T;
#pragma unroll
        for (int i = 0; i < PFD; ++i) {
            const int c = tid + P::NT * i;
            if (c < P::NTOK * P::CPT) *reinterpret_cast<uint4*>(dst + c * 16) = (fok && doff[i] >= 0) ? pdc[i] : make_uint4(0, 0, 0, 0);
        }
    };

    float acc[27][2];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t][0] = acc[t][1] = 0.f;

    // input frames [f0, f1) of this workgroup (gridDim.z slices of T: small grids are split to fill the chip)
    const int fz = (T + (int)gridDim.z - 1) / (int)gridDim.z;
    const int f0 = (int)blockIdx.z * fz, f1 = f0 + fz < T ? f0 + fz : T;
    if (f0 < f1) {
        pre_in(f0);
        pre_dc(f0 - 1);
        commit_in();
        commit_dc(f0 - 1);                     // zero-filled outside [0, T)
        pre_dc(f0);
        commit_dc(f0);
        pre_dc(f0 + 1);
        commit_dc(f0 + 1);
    }
    __syncthreads();
    for (int f = f0; f < f1; ++f) {
        if (f + 1 < f1) {                      // next frame's tiles: in flight under this frame's arithmetic
            pre_in(f + 1);
            pre_dc(f + 2);
        }
        // d_conv rows of this thread for output frames f+1, f, f-1  (tap dt = 0, 1, 2)
        float d[3][P::XO][2];
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
            const int fo = f + 1 - dt;
            const bool ok = fo >= 0 && fo < T;
            const TA* dp = reinterpret_cast<const TA*>(dc_lds + ((fo + 3) % 3) * DT_BYTES) + (row * P::XO) * 96 + 2 * cp;
#pragma unroll
            for (int x = 0; x < P::XO; ++x) {
                d[dt][x][0] = 0.f; d[dt][x][1] = 0.f;
                if (ok) load2<TA>(dp + x * 96, d[dt][x][0], d[dt][x][1]);
            }
        }
        const TA* tile = reinterpret_cast<const TA*>(in_lds) + 2 * cp;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            float xin[P::IW][2];
            const TA* rp = tile + (S * row + dy) * P::IW * 96;
#pragma unroll
            for (int ix = 0; ix < P::IW; ++ix) load2<TA>(rp + ix * 96, xin[ix][0], xin[ix][1]);
#pragma unroll
            for (int dt = 0; dt < 3; ++dt)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    float a0 = acc[(dt * 3 + dy) * 3 + dx][0], a1 = acc[(dt * 3 + dy) * 3 + dx][1];
#pragma unroll
                    for (int x = 0; x < P::XO; ++x) {
                        a0 = fmaf(d[dt][x][0], xin[S * x + dx][0], a0);
                        a1 = fmaf(d[dt][x][1], xin[S * x + dx][1], a1);
                    }
                    acc[(dt * 3 + dy) * 3 + dx][0] = a0;
                    acc[(dt * 3 + dy) * 3 + dx][1] = a1;
                }
        }
        __syncthreads();                       // everyone is done with the input tile and with d_conv frame f-1
        if (f + 1 < f1) {
            commit_in();
            commit_dc(f + 2);                  // slot (f+2)%3 == (f-1)%3; zero-filled when f+2 >= T
        }
        __syncthreads();
    }
    // combine the ROWS threads that share a channel pair in ROW ORDER (a fixed summation order: LDS float atomics would add in
    // arrival order and make the weight gradient differ in its last bits from run to run): the rows park their 54 sums side by
    // side, four rows per pass (4 x 2592 floats fit the tile storage this aliases), and every output is added up in row order
    float* red = reinterpret_cast<float*>(smem);   // [4][2592]
    static_assert(4 * 2592 * 4 <= P::IN_BYTES + 3 * P::NTOK * 96 * (int)sizeof(TA), "row slabs must fit");
    constexpr int NOUT = (2592 + P::NT - 1) / P::NT;
    float tot[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) tot[o] = 0.f;
    for (int r0 = 0; r0 < P::ROWS; r0 += 4) {
        __syncthreads();                           // everyone is done with what this aliases (tiles / the previous pass)
        if (row >= r0 && row < r0 + 4) {
            // parked as [tap][channel] with the channel PAIR of a thread adjacent: one 8-byte store per tap from the register pair
            // the arithmetic keeps together (a [channel][tap] image lets the compiler merge stores across taps, which forces the
            // 54 sums into consecutive registers for the whole kernel: 150 spilled registers in the main loop)
#pragma unroll
            for (int t = 0; t < 27; ++t)
                *reinterpret_cast<float2*>(&red[(row - r0) * 2592 + t * 96 + 2 * cp]) = make_float2(acc[t][0], acc[t][1]);
        }
        __syncthreads();
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            const int i = tid + P::NT * o;          // output index c * 27 + tap
            if (i < 2592) {
                const int c = i / 27, t = i - c * 27;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
                    if (r0 + rr < P::ROWS) tot[o] += red[rr * 2592 + t * 96 + c];
            }
        }
    }
    float* prow = part + (((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 2592;
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        const int i = tid + P::NT * o;
        if (i < 2592) prow[i] = tot[o];
    }
}

template <typename TA, int S>
static int launch_pool_wgrad_tiled(const void* qkv, int64_t ld, int chan_off, const void* dconv, float* part, int B, int heads,
                                   int T, int H, int W, int Ho, int Wo, hipStream_t st, int nset = 1) {
    using P = PoolTile<TA, S>;
    constexpr int SM = P::IN_BYTES + 3 * P::NTOK * 96 * (int)sizeof(TA);
    static_assert(SM >= 2592 * 4, "partial row must fit");
    dim3 grid(((Wo + P::XO - 1) / P::XO) * ((Ho + P::ROWS - 1) / P::ROWS), nset * B * heads);
    static DevFlags attr_done_tab; DevFlag attr_done = dev_flag(attr_done_tab);
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pool_wgrad_tiled_kernel<TA, S>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, SM) != hipSuccess)
            return MVIT_ELAUNCH;
        attr_done = true;
    }
    // (a gridDim.z split of the frames is supported by the kernel but measured slower here: the prologue and the LDS reduction of
    // the 2592 partial sums are paid per workgroup)
    hipLaunchKernelGGL((pool_wgrad_tiled_kernel<TA, S>), grid, dim3(P::NT), SM, st, (const TA*)qkv, ld, chan_off, (const TA*)dconv,
                       part, heads, T, H, W, Ho, Wo, nset == 2 ? B * heads : 0);
    MVIT_LAUNCH_CHECK();
    return (int)(grid.x * grid.y * grid.z);
}

// ------------------------------------------------------------------------------------------------
// Data gradient of the STRIDE-2 pooling conv, tiled.  dX[t][y][x] = sum over taps with (y+1-dy), (x+1-dx) even of
// W[dt][dy][dx] * dC[t+1-dt][(y+1-dy)/2][(x+1-dx)/2]: the four parities of (y, x) are small convolutions over the d_conv grid
// (1, 2, 2 and 4 spatial taps), so one workgroup takes a 4 x 8 block of d_conv cells (+1 halo row / column), keeps the d_conv
// tiles of three output frames in an LDS ring and produces the 8 x 16 input tokens of every input frame from it; thread =
// (channel pair, cell row).  Results are staged as packed 16-bit pairs and written as whole 192-byte token rows.
// ------------------------------------------------------------------------------------------------
template <typename TA>
struct Dgrad2Tile {
    static constexpr int ROWS = 4, XO = 8;
    static constexpr int CH = ROWS + 1, CWD = XO + 1;                 // d_conv cells incl. halo
    static constexpr int NT = 48 * ROWS;                              // 192 threads
    static constexpr int CW = 16 / sizeof(TA), CPT = 96 / CW;
    static constexpr int DC_CHUNKS = CH * CWD * CPT;
    static constexpr int PF = (DC_CHUNKS + NT - 1) / NT;
    static constexpr int DC_BYTES = CH * CWD * 96 * (int)sizeof(TA);
    static constexpr int NTOK = 2 * ROWS * 2 * XO;                    // 128 input tokens per frame
    static constexpr int ST_BYTES = NTOK * 96 * (int)sizeof(TA);      // staged results, act-typed
    static constexpr int W_BYTES = 27 * 96 * 4;
    static constexpr int SMEM = 3 * DC_BYTES + ST_BYTES + W_BYTES;
};

template <typename TA>
__global__ __launch_bounds__(192, 2) void pool_dgrad2_tiled_kernel(const TA* __restrict__ dconv, const float* __restrict__ w,
                                                                   TA* __restrict__ dqkv, int64_t ld, int chan_off, int heads, int T,
                                                                   int H, int W, int Ho, int Wo, int set_bh = 0,
                                                                   const float* __restrict__ w2 = nullptr) {
    using P = Dgrad2Tile<TA>;
    constexpr int CW = P::CW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* dc_lds = smem;                                              // [3][CH*CWD][96]
    TA* stage = reinterpret_cast<TA*>(smem + 3 * P::DC_BYTES);        // [NTOK][96]
    float* wl = reinterpret_cast<float*>(smem + 3 * P::DC_BYTES + P::ST_BYTES);   // [tap][channel]
    const int tid = threadIdx.x;
    const int cp = tid % 48, row = tid / 48;
    const int tiles_x = (Wo + P::XO - 1) / P::XO;
    const int xo0 = (blockIdx.x % tiles_x) * P::XO, yo0 = (blockIdx.x / tiles_x) * P::ROWS;
    const int bh = blockIdx.y;                  // set_bh > 0: two tensors in one launch (see pool_tiled_kernel)
    int bhs = bh, hoff = 0;
    if (set_bh > 0 && bh >= set_bh) { bhs = bh - set_bh; hoff = heads; w = w2; }
    const int b = bhs / heads, g = hoff + (bhs - b * heads);          // g: head index inside the fused buffer's channel slice
    const TA* dbase = dconv + (int64_t)bh * T * Ho * Wo * 96;
    for (int i = tid; i < 27 * 96; i += P::NT) {
        const int tap = i / 96, c = i - tap * 96;
        wl[i] = w[c * 27 + tap];
    }
    // register prefetch of one d_conv frame tile (unconditional clamped loads; cells outside the grid are zeroed at commit)
    uint4 pf[P::PF];
    int doff[P::PF];
#pragma unroll
    for (int i = 0; i < P::PF; ++i) {
        const int c = tid + P::NT * i;
        doff[i] = -1;
        if (c < P::DC_CHUNKS) {
            const int cell = c / P::CPT, ch = c - cell * P::CPT;
            const int yo = yo0 + cell / P::CWD, xo = xo0 + cell % P::CWD;
            if (yo < Ho && xo < Wo) doff[i] = (yo * Wo + xo) * 96 + ch * CW;
        }
    }
    const int64_t dc_frame = (int64_t)Ho * Wo * 96;
    auto prefetch = [&](int fo) {
        const TA* fb = dbase + (fo >= 0 && fo < T ? fo : 0) * dc_frame;
#pragma unroll
        for (int i = 0; i < P::PF; ++i) pf[i] = *reinterpret_cast<const uint4*>(fb + (doff[i] >= 0 ? doff[i] : 0));
    };
    auto commit = [&](int fo) {     // frame fo -> ring slot (fo + 3) % 3; zeros outside [0, T) and outside the grid
        char* dst = dc_lds + ((fo + 3) % 3) * P::DC_BYTES;
        const bool fok = fo >= 0 && fo < T;
#pragma unroll
        for (int i = 0; i < P::PF; ++i) {
            const int c = tid + P::NT * i;
            if (c < P::DC_CHUNKS) *reinterpret_cast<uint4*>(dst + c * 16) = (fok && doff[i] >= 0) ? pf[i] : make_uint4(0, 0, 0, 0);
        }
    };
    // input frame t needs d_conv frames t+1 (dt=0), t (dt=1), t-1 (dt=2)
    const int tz = (T + (int)gridDim.z - 1) / (int)gridDim.z;     // input frames [t0, t1) of this workgroup
    const int t0 = (int)blockIdx.z * tz, t1 = t0 + tz < T ? t0 + tz : T;
    prefetch(t0 - 1); commit(t0 - 1);
    prefetch(t0);     commit(t0);
    prefetch(t0 + 1); commit(t0 + 1);
    __syncthreads();
    const float* wm0 = wl + 2 * cp;
    const int64_t Nin = (int64_t)T * H * W;
    for (int t = t0; t < t1; ++t) {
        if (t + 1 < t1) prefetch(t + 2);         // lands under this frame's arithmetic (frame t+2 replaces t-1 after the barrier)
        float acc[2][2 * P::XO][2];              // [input row parity][x within the 16][channel]
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int x = 0; x < 2 * P::XO; ++x) acc[py][x][0] = acc[py][x][1] = 0.f;
        const lds_cptr_t wm = lds_opaque(wm0);    // keep the weight reads inside the loop, as LDS reads
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
            const TA* tile = reinterpret_cast<const TA*>(dc_lds + ((t + 1 - dt + 3) % 3) * P::DC_BYTES) + 2 * cp;
            float d0[P::CWD][2], d1[P::CWD][2];  // d_conv rows yo (= row) and yo + 1
#pragma unroll
            for (int x = 0; x < P::CWD; ++x) {
                load2<TA>(tile + (row * P::CWD + x) * 96, d0[x][0], d0[x][1]);
                load2<TA>(tile + ((row + 1) * P::CWD + x) * 96, d1[x][0], d1[x][1]);
            }
            // weights of this dt: wv[dy][dx]
            float2 wv[3][3];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const lds_f32x2_t t2 = lds_ld<lds_f32x2_t>(wm, ((dt * 3 + dy) * 3 + dx) * 96 * 4);
                    wv[dy][dx] = make_float2(t2.x, t2.y);
                }
#pragma unroll
            for (int xo = 0; xo < P::XO; ++xo) {
                // even input row y = 2*yo: dy = 1 -> d_conv row yo;  odd row y = 2*yo+1: dy = 0 -> row yo+1, dy = 2 -> row yo
                // even x = 2*xo: dx = 1 -> col xo;  odd x = 2*xo+1: dx = 0 -> col xo+1, dx = 2 -> col xo
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float w11 = c ? wv[1][1].y : wv[1][1].x, w10 = c ? wv[1][0].y : wv[1][0].x, w12 = c ? wv[1][2].y : wv[1][2].x;
                    const float w01 = c ? wv[0][1].y : wv[0][1].x, w21 = c ? wv[2][1].y : wv[2][1].x;
                    const float w00 = c ? wv[0][0].y : wv[0][0].x, w02 = c ? wv[0][2].y : wv[0][2].x;
                    const float w20 = c ? wv[2][0].y : wv[2][0].x, w22 = c ? wv[2][2].y : wv[2][2].x;
                    const float a = d0[xo][c], bq = d0[xo + 1][c], cq = d1[xo][c], dq = d1[xo + 1][c];
                    acc[0][2 * xo][c] = fmaf(w11, a, acc[0][2 * xo][c]);
                    acc[0][2 * xo + 1][c] = fmaf(w10, bq, fmaf(w12, a, acc[0][2 * xo + 1][c]));
                    acc[1][2 * xo][c] = fmaf(w01, cq, fmaf(w21, a, acc[1][2 * xo][c]));
                    acc[1][2 * xo + 1][c] = fmaf(w00, dq, fmaf(w02, cq, fmaf(w20, bq, fmaf(w22, a, acc[1][2 * xo + 1][c]))));
                }
            }
        }
        // stage the 2 x 16 tokens of this thread (channel pair cp) and write whole token rows
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int x = 0; x < 2 * P::XO; ++x) {
                const int tok = (2 * row + py) * (2 * P::XO) + x;
                store2<TA>(stage + tok * 96 + 2 * cp, acc[py][x][0], acc[py][x][1]);
            }
        __syncthreads();
        for (int c = tid; c < P::NTOK * P::CPT; c += P::NT) {
            const int tok = c / P::CPT, ch = c - tok * P::CPT;
            const int y = 2 * yo0 + tok / (2 * P::XO), x = 2 * xo0 + tok % (2 * P::XO);
            if (y < H && x < W)
                *reinterpret_cast<uint4*>(dqkv + ((int64_t)b * Nin + ((int64_t)t * H + y) * W + x) * ld + chan_off + g * 96 + ch * CW) =
                    *reinterpret_cast<const uint4*>(stage + tok * 96 + ch * CW);
        }
        if (t + 1 < t1) commit(t + 2);           // slot of frame t-1: every thread is past its reads (barrier above)
        __syncthreads();
    }
}

template <typename TA>
static int launch_pool_dgrad2_tiled(const void* dconv, const float* w, void* dqkv, int64_t ld, int chan_off, int B, int heads, int T,
                                    int H, int W, int Ho, int Wo, hipStream_t st, const float* w2 = nullptr) {
    using P = Dgrad2Tile<TA>;
    const int nset = w2 ? 2 : 1;
    dim3 grid(((Wo + P::XO - 1) / P::XO) * ((Ho + P::ROWS - 1) / P::ROWS), nset * B * heads);
    static DevFlags attr_done_tab; DevFlag attr_done = dev_flag(attr_done_tab);
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pool_dgrad2_tiled_kernel<TA>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                P::SMEM) != hipSuccess)
            return MVIT_ELAUNCH;
        attr_done = true;
    }
    if (grid.x * grid.y * 2 <= 512 && T >= 4) grid.z = 2;        // small grids: split the frames to fill the chip
    hipLaunchKernelGGL((pool_dgrad2_tiled_kernel<TA>), grid, dim3(P::NT), P::SMEM, st, (const TA*)dconv, w, (TA*)dqkv, ld, chan_off, heads,
                       T, H, W, Ho, Wo, nset == 2 ? B * heads : 0, w2);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}
int mvit_internal_pool_dgrad2_tiled(const void* dconv, const float* w, void* dqkv, int64_t ld, int chan_off, int B, int heads, int T,
                                    int H, int W, int act_dtype, hipStream_t st) {
    if ((int64_t)B * heads > 65535) return MVIT_EINVAL;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    return act_dtype == MVIT_BF16 ? launch_pool_dgrad2_tiled<bf16_t>(dconv, w, dqkv, ld, chan_off, B, heads, T, H, W, Ho, Wo, st)
                                  : launch_pool_dgrad2_tiled<float>(dconv, w, dqkv, ld, chan_off, B, heads, T, H, W, Ho, Wo, st);
}

// internal: data gradient of the stride-1 pooling conv through the tiled kernel (PLAIN mode); dconv [B*heads][T*H*W][96] ->
// the (chan_off) slice of dqkv [B][T*H*W][ld]
template <typename TA>
static int launch_pool_dgrad_tiled(const void* dconv, const float* w, void* dqkv, int64_t ld, int chan_off, int B, int heads, int T,
                                   int H, int W, hipStream_t st) {
    using P = PoolTile<TA, 1>;
    dim3 grid(((W + P::XO - 1) / P::XO) * ((H + P::ROWS - 1) / P::ROWS), B * heads);
    static DevFlags attr_done_tab; DevFlag attr_done = dev_flag(attr_done_tab);
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pool_tiled_kernel<TA, 1, false, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, P::SMEM) != hipSuccess)
            return MVIT_ELAUNCH;
        attr_done = true;
    }
    hipLaunchKernelGGL((pool_tiled_kernel<TA, 1, false, true>), grid, dim3(P::NT), P::SMEM, st, (const TA*)dconv, (int64_t)96, 0, w,
                       (const float*)nullptr, (const float*)nullptr, (TA*)dqkv, (const TA*)nullptr, (float*)nullptr, 1, T, H, W, H, W,
                       0.f, ld, chan_off, heads);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}
int mvit_internal_pool_march_dgrad1(const void* dconv, const float* w, void* dqkv, int64_t ld, int chan_off, int B, int heads, int T,
                                    int H, int W, int act_dtype, hipStream_t st);
int mvit_internal_pool_dgrad_tiled(const void* dconv, const float* w, void* dqkv, int64_t ld, int chan_off, int B, int heads, int T,
                                   int H, int W, int act_dtype, hipStream_t st) {
    if ((int64_t)B * heads > 65535) return MVIT_EINVAL;
    static const int march_env = getenv("MVIT_POOL_MARCH") ? atoi(getenv("MVIT_POOL_MARCH")) : -1;
    if (march_env >= 0 ? march_env != 0 : W <= 56) {
        const int rc = mvit_internal_pool_march_dgrad1(dconv, w, dqkv, ld, chan_off, B, heads, T, H, W, act_dtype, st);
        if (rc != MVIT_EUNSUPPORTED) return rc;
    }
    return act_dtype == MVIT_BF16 ? launch_pool_dgrad_tiled<bf16_t>(dconv, w, dqkv, ld, chan_off, B, heads, T, H, W, st)
                                  : launch_pool_dgrad_tiled<float>(dconv, w, dqkv, ld, chan_off, B, heads, T, H, W, st);
}

// internal: returns the number of [2592]-float partial rows written, or a negative error (strides 1 and 2 only)
// the k and v tensors of a block in one launch each (stride 2; dconv / outputs of the two sets back to back, see pool_tiled_kernel)
int mvit_internal_pool_dgrad2_tiled_kv(const void* dconv_kv, const float* w_k, const float* w_v, void* dqkv, int64_t ld, int chan_off_k,
                                       int B, int heads, int T, int H, int W, int act_dtype, hipStream_t st) {
    if ((int64_t)2 * B * heads > 65535) return MVIT_EINVAL;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    return act_dtype == MVIT_BF16 ? launch_pool_dgrad2_tiled<bf16_t>(dconv_kv, w_k, dqkv, ld, chan_off_k, B, heads, T, H, W, Ho, Wo, st, w_v)
                                  : launch_pool_dgrad2_tiled<float>(dconv_kv, w_k, dqkv, ld, chan_off_k, B, heads, T, H, W, Ho, Wo, st, w_v);
}
// pool_march.hip: the march form of the weight gradient (16-bit builds: v_dot2c over token pairs, 7 x 7 tiles); MVIT_POOL_WGRAD_MARCH=0
// keeps the 8-wide tiles below (A/B)
int mvit_internal_pool_wgrad_march(const void* qkv, int64_t ld, int chan_off, const void* dconv, float* part, int B, int heads, int T, int H,
                                   int W, int stride_hw, int nset, hipStream_t st, const void* xhat = nullptr, const void* dout = nullptr,
                                   const float* rstd = nullptr, const float* gamma = nullptr, const float* gamma2 = nullptr,
                                   float* part_ln = nullptr);
static bool wgrad_march_on() {
    static const bool on = !(getenv("MVIT_POOL_WGRAD_MARCH") && getenv("MVIT_POOL_WGRAD_MARCH")[0] == '0');
    return on;
}
int mvit_internal_pool_wgrad_tiled_kv(const void* qkv, int64_t ld, int chan_off_k, const void* dconv_kv, float* part, int B, int heads,
                                      int T, int H, int W, int act_dtype, hipStream_t st) {       // returns the partial rows of BOTH sets
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    if (act_dtype == MVIT_BF16 && wgrad_march_on()) {
        const int rc = mvit_internal_pool_wgrad_march(qkv, ld, chan_off_k, dconv_kv, part, B, heads, T, H, W, 2, 2, st);
        if (rc != MVIT_EUNSUPPORTED) return rc;
    }
    return act_dtype == MVIT_BF16 ? launch_pool_wgrad_tiled<bf16_t, 2>(qkv, ld, chan_off_k, dconv_kv, part, B, heads, T, H, W, Ho, Wo, st, 2)
                                  : launch_pool_wgrad_tiled<float, 2>(qkv, ld, chan_off_k, dconv_kv, part, B, heads, T, H, W, Ho, Wo, st, 2);
}
int mvit_internal_pool_wgrad_tiled(const void* qkv, int64_t ld, int chan_off, const void* dconv, float* part, int B, int heads,
                                   int T, int H, int W, int stride_hw, int act_dtype, hipStream_t st) {
    const int Ho = (H - 1) / stride_hw + 1, Wo = (W - 1) / stride_hw + 1;
    if (act_dtype == MVIT_BF16 && wgrad_march_on()) {
        const int rc = mvit_internal_pool_wgrad_march(qkv, ld, chan_off, dconv, part, B, heads, T, H, W, stride_hw, 1, st);
        if (rc != MVIT_EUNSUPPORTED) return rc;
    }
    if (stride_hw == 1)
        return act_dtype == MVIT_BF16 ? launch_pool_wgrad_tiled<bf16_t, 1>(qkv, ld, chan_off, dconv, part, B, heads, T, H, W, Ho, Wo, st)
                                      : launch_pool_wgrad_tiled<float, 1>(qkv, ld, chan_off, dconv, part, B, heads, T, H, W, Ho, Wo, st);
    return act_dtype == MVIT_BF16 ? launch_pool_wgrad_tiled<bf16_t, 2>(qkv, ld, chan_off, dconv, part, B, heads, T, H, W, Ho, Wo, st)
                                  : launch_pool_wgrad_tiled<float, 2>(qkv, ld, chan_off, dconv, part, B, heads, T, H, W, Ho, Wo, st);
}

// internal (used by pool_bwd.hip): LN-backward of the pooled conv through the tiled march; returns the number of partial rows
// written to part ([rows][96], d_gamma only) or a negative error.  Only strides 1 and 2.
int mvit_internal_pool_ln_bwd_tiled(const void* qkv, int64_t ld, int chan_off, const float* w, const float* gamma, const void* dout,
                                    void* dconv, float* part, int B, int heads, int T, int H, int W, int stride_hw, float eps,
                                    int act_dtype, hipStream_t st) {
    const int Ho = (H - 1) / stride_hw + 1, Wo = (W - 1) / stride_hw + 1;
    int rc;
    int rows;
    if (stride_hw == 1) {
        rows = ((Wo + 7) / 8) * ((Ho + 7) / 8) * B * heads;
        rc = act_dtype == MVIT_BF16 ? launch_pool_tiled_bwd<bf16_t, 1>(qkv, ld, chan_off, w, gamma, dout, dconv, part, B, heads, T, H, W, Ho, Wo, eps, st)
                                    : launch_pool_tiled_bwd<float, 1>(qkv, ld, chan_off, w, gamma, dout, dconv, part, B, heads, T, H, W, Ho, Wo, eps, st);
    } else {
        rows = ((Wo + 7) / 8) * ((Ho + 3) / 4) * B * heads;
        rc = act_dtype == MVIT_BF16 ? launch_pool_tiled_bwd<bf16_t, 2>(qkv, ld, chan_off, w, gamma, dout, dconv, part, B, heads, T, H, W, Ho, Wo, eps, st)
                                    : launch_pool_tiled_bwd<float, 2>(qkv, ld, chan_off, w, gamma, dout, dconv, part, B, heads, T, H, W, Ho, Wo, eps, st);
    }
    return rc < 0 ? rc : rows;
}

int mvit_internal_pool_march_fwd(const void* qkv, int64_t ld, int chan_off, const float* w, const float* gamma, const float* beta, void* out,
                                 void* xhat, float* rstd, int B, int heads, int T, int H, int W, int stride_hw, float eps, int act_dtype,
                                 hipStream_t st, const float* w2 = nullptr, const float* gamma2 = nullptr,
                                 const float* beta2 = nullptr);      // pool_march.hip
int mvit_internal_pool_march_dgrad1(const void* dconv, const float* w, void* dqkv, int64_t ld, int chan_off, int B, int heads, int T,
                                    int H, int W, int act_dtype, hipStream_t st);

// Training forward: additionally keeps xhat = (conv - mean) * rstd ([B][heads][T*Ho*Wo][96], act-typed) and rstd (fp32 per token),
// which is all the LayerNorm backward needs -- the backward then skips the second convolution (mvit_pool_conv_ln_bwd_saved).
extern "C" int mvit_pool_conv_ln_fwd_train(const void* qkv, int64_t ld, int chan_off, const float* w, const float* gamma,
                                           const float* beta, void* out, void* xhat, float* rstd, int B, int heads, int T, int H,
                                           int W, int stride_hw, float eps, int act_dtype, void* stream) {
    if (!qkv || !w || !gamma || !beta || !out || (xhat && !rstd) || B <= 0 || heads <= 0 || T <= 0 || H <= 0 || W <= 0 || stride_hw <= 0)
        return MVIT_EINVAL;
    if ((ld & 7) || (chan_off & 7)) return MVIT_EUNSUPPORTED;
    const int Ho = (H - 1) / stride_hw + 1, Wo = (W - 1) / stride_hw + 1;
    const int64_t total = (int64_t)B * heads * T * Ho * Wo;
    int64_t blocks = (total + 63) / 64;
    if (blocks > 16384) blocks = 16384;
    hipStream_t st = as_stream(stream);
    if (act_dtype != MVIT_F32 && act_dtype != MVIT_BF16) return MVIT_EDTYPE;
    if ((int64_t)B * heads > 65535) return MVIT_EINVAL;
    if (stride_hw == 1 || stride_hw == 2) {
        // pool_march.hip (7 x 7 tiles, in-register LayerNorm).  Rounds 2-4 used it for stride 1 on the <= 56 x 56 grids only; since the
        // round-5 work on it (LDS weight reads, static accumulator sets, transposed LayerNorm reduction) it is ahead of the 8-wide tiles
        // below on every shape of the model (profiles/r5_pool_*.txt): stride 2 at 28 / 56 / 112: 25.6 / 28.5 / 60.8 against 29.1 / 36.0 /
        // 70.1 us.  MVIT_POOL_MARCH=0 / 1 forces one form (A/B); MVIT_POOL_MARCH_MAXW bounds the grids it takes at stride 1.
        static const int march_env = getenv("MVIT_POOL_MARCH") ? atoi(getenv("MVIT_POOL_MARCH")) : -1;
        static const int march_maxw = getenv("MVIT_POOL_MARCH_MAXW") ? atoi(getenv("MVIT_POOL_MARCH_MAXW")) : (1 << 30);    // (112 x 112: 158 vs 172 us)
        const bool march = march_env >= 0 ? march_env != 0 : (stride_hw == 2 || W <= march_maxw);
        if (march) {
            const int rc = mvit_internal_pool_march_fwd(qkv, ld, chan_off, w, gamma, beta, out, xhat, rstd, B, heads, T, H, W, stride_hw, eps,
                                                        act_dtype, st);
            if (rc != MVIT_EUNSUPPORTED) return rc;
        }
        if (act_dtype == MVIT_BF16) {
            if (stride_hw == 1) return launch_pool_tiled<bf16_t, 1>(qkv, ld, chan_off, w, gamma, beta, out, xhat, rstd, B, heads, T, H, W, Ho, Wo, eps, st);
            return launch_pool_tiled<bf16_t, 2>(qkv, ld, chan_off, w, gamma, beta, out, xhat, rstd, B, heads, T, H, W, Ho, Wo, eps, st);
        }
        if (stride_hw == 1) return launch_pool_tiled<float, 1>(qkv, ld, chan_off, w, gamma, beta, out, xhat, rstd, B, heads, T, H, W, Ho, Wo, eps, st);
        return launch_pool_tiled<float, 2>(qkv, ld, chan_off, w, gamma, beta, out, xhat, rstd, B, heads, T, H, W, Ho, Wo, eps, st);
    }
    if (act_dtype == MVIT_F32)
        hipLaunchKernelGGL((pool_conv_ln_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, st, (const float*)qkv, ld,
                           chan_off, w, gamma, beta, (float*)out, (float*)xhat, rstd, B, heads, T, H, W, Ho, Wo, stride_hw, eps);
    else if (act_dtype == MVIT_BF16)
        hipLaunchKernelGGL((pool_conv_ln_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)qkv, ld,
                           chan_off, w, gamma, beta, (bf16_t*)out, (bf16_t*)xhat, rstd, B, heads, T, H, W, Ho, Wo, stride_hw, eps);
    else
        return MVIT_EDTYPE;
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}


// k and v pooling conv + LayerNorm of a block in ONE launch (stride 2 only; other strides: MVIT_EUNSUPPORTED, call the single form
// twice).  The v head group follows the k group in the fused qkv buffer (chan_off_k + heads * 96); out_kv / xhat_kv are
// [2][B][heads][T*Ho*Wo][96] (k then v), rstd_kv [2][B*heads*T*Ho*Wo].
extern "C" int mvit_pool_conv_ln_fwd_train_kv(const void* qkv, int64_t ld, int chan_off_k, const float* w_k, const float* gamma_k,
                                              const float* beta_k, const float* w_v, const float* gamma_v, const float* beta_v,
                                              void* out_kv, void* xhat_kv, float* rstd_kv, int B, int heads, int T, int H, int W,
                                              int stride_hw, float eps, int act_dtype, void* stream) {
    if (!qkv || !w_k || !gamma_k || !beta_k || !w_v || !gamma_v || !beta_v || !out_kv || B <= 0 || heads <= 0 || T <= 0 || H <= 0 || W <= 0)
        return MVIT_EINVAL;
    if ((xhat_kv == nullptr) != (rstd_kv == nullptr)) return MVIT_EINVAL;
    if (act_dtype != MVIT_F32 && act_dtype != MVIT_BF16) return MVIT_EDTYPE;
    if ((ld & 7) || (chan_off_k & 7)) return MVIT_EUNSUPPORTED;
    if (stride_hw != 2 || (int64_t)2 * B * heads > 65535) return MVIT_EUNSUPPORTED;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    hipStream_t st = as_stream(stream);
    static const int march_env = getenv("MVIT_POOL_MARCH") ? atoi(getenv("MVIT_POOL_MARCH")) : -1;
    if (march_env != 0) {       // the pair form of the march kernel (pool_march.hip): 2 x B x heads x tiles workgroups in one launch
        const int rc = mvit_internal_pool_march_fwd(qkv, ld, chan_off_k, w_k, gamma_k, beta_k, out_kv, xhat_kv, rstd_kv, B, heads, T, H, W, 2, eps,
                                                    act_dtype, st, w_v, gamma_v, beta_v);
        if (rc != MVIT_EUNSUPPORTED) return rc;
    }
    if (act_dtype == MVIT_BF16)
        return launch_pool_tiled<bf16_t, 2>(qkv, ld, chan_off_k, w_k, gamma_k, beta_k, out_kv, xhat_kv, rstd_kv, B, heads, T, H, W, Ho, Wo, eps, st,
                                            w_v, gamma_v, beta_v);
    return launch_pool_tiled<float, 2>(qkv, ld, chan_off_k, w_k, gamma_k, beta_k, out_kv, xhat_kv, rstd_kv, B, heads, T, H, W, Ho, Wo, eps, st, w_v,
                                       gamma_v, beta_v);
}

extern "C" int mvit_pool_conv_ln_fwd(const void* qkv, int64_t ld, int chan_off, const float* w, const float* gamma,
                                     const float* beta, void* out, int B, int heads, int T, int H, int W, int stride_hw,
                                     float eps, int act_dtype, void* stream) {
    return mvit_pool_conv_ln_fwd_train(qkv, ld, chan_off, w, gamma, beta, out, nullptr, nullptr, B, heads, T, H, W, stride_hw, eps,
                                       act_dtype, stream);
}
