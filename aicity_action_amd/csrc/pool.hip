// Pooling conv (depthwise 3x3x3, stride (1,s,s), zero pad 1, weight shared across heads) fused with the
// LayerNorm(96, eps) that follows it; reads the fused qkv activation in place (no head-split copies).
// HBM/L2-bound stencil: 4 lanes per output token, 24 channels per lane as 16-byte chunks
// (chunk index = lane + 4*i, so one wave-instruction reads 64 contiguous bytes per token).
#include "common.h"

template <typename TA>
__global__ __launch_bounds__(256) void pool_conv_ln_kernel(const TA* __restrict__ qkv, int64_t ld, int chan_off,
                                                           const float* __restrict__ w, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, TA* __restrict__ out, int B,
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
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int xi = xo * s + dx - 1;
                    if (xi < 0 || xi >= W) continue;
                    const TA* p = base + (((int64_t)ti * H + yi) * W + xi) * ld;
                    const float* wt = wsm + ((dt * 3 + dy) * 3 + dx) * 96;
#pragma unroll
                    for (int i = 0; i < NCH; ++i) {
                        const int c0 = CW * (j + 4 * i);
                        if constexpr (sizeof(TA) == 2) {
                            float4 lo, hi;
                            load8(p + c0, lo, hi);
                            const float4 w0 = *reinterpret_cast<const float4*>(wt + c0);
                            const float4 w1 = *reinterpret_cast<const float4*>(wt + c0 + 4);
                            acc[i * 8 + 0] = fmaf(lo.x, w0.x, acc[i * 8 + 0]);
                            acc[i * 8 + 1] = fmaf(lo.y, w0.y, acc[i * 8 + 1]);
                            acc[i * 8 + 2] = fmaf(lo.z, w0.z, acc[i * 8 + 2]);
                            acc[i * 8 + 3] = fmaf(lo.w, w0.w, acc[i * 8 + 3]);
                            acc[i * 8 + 4] = fmaf(hi.x, w1.x, acc[i * 8 + 4]);
                            acc[i * 8 + 5] = fmaf(hi.y, w1.y, acc[i * 8 + 5]);
                            acc[i * 8 + 6] = fmaf(hi.z, w1.z, acc[i * 8 + 6]);
                            acc[i * 8 + 7] = fmaf(hi.w, w1.w, acc[i * 8 + 7]);
                        } else {
                            const float4 v = load4(p + c0);
                            const float4 w0 = *reinterpret_cast<const float4*>(wt + c0);
                            acc[i * 4 + 0] = fmaf(v.x, w0.x, acc[i * 4 + 0]);
                            acc[i * 4 + 1] = fmaf(v.y, w0.y, acc[i * 4 + 1]);
                            acc[i * 4 + 2] = fmaf(v.z, w0.z, acc[i * 4 + 2]);
                            acc[i * 4 + 3] = fmaf(v.w, w0.w, acc[i * 4 + 3]);
                        }
                    }
                }
            }
        }
        // LayerNorm over the 96 channels of this token (4 lanes x 24)
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

extern "C" int mvit_pool_conv_ln_fwd(const void* qkv, int64_t ld, int chan_off, const float* w, const float* gamma,
                                     const float* beta, void* out, int B, int heads, int T, int H, int W, int stride_hw,
                                     float eps, int act_dtype, void* stream) {
    if (!qkv || !w || !gamma || !beta || !out || B <= 0 || heads <= 0 || T <= 0 || H <= 0 || W <= 0 || stride_hw <= 0)
        return MVIT_EINVAL;
    if ((ld & 7) || (chan_off & 7)) return MVIT_EUNSUPPORTED;
    const int Ho = (H - 1) / stride_hw + 1, Wo = (W - 1) / stride_hw + 1;
    const int64_t total = (int64_t)B * heads * T * Ho * Wo;
    int64_t blocks = (total + 63) / 64;
    if (blocks > 16384) blocks = 16384;
    hipStream_t st = as_stream(stream);
    if (act_dtype == MVIT_F32)
        hipLaunchKernelGGL((pool_conv_ln_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, st, (const float*)qkv, ld,
                           chan_off, w, gamma, beta, (float*)out, B, heads, T, H, W, Ho, Wo, stride_hw, eps);
    else if (act_dtype == MVIT_BF16)
        hipLaunchKernelGGL((pool_conv_ln_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)qkv, ld,
                           chan_off, w, gamma, beta, (bf16_t*)out, B, heads, T, H, W, Ho, Wo, stride_hw, eps);
    else
        return MVIT_EDTYPE;
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}
