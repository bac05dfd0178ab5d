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

extern "C" int mvit_stem_fwd(const float* clip, const float* w, const float* bias, const float* pos_spatial,
                             const float* pos_temporal, float* x, int B, int T, int S, int act_dtype, void* stream) {
    if (!clip || !w || !bias || !pos_spatial || !pos_temporal || !x || B <= 0 || T <= 0 || S <= 0) return MVIT_EINVAL;
    if ((T & 1) || (S & 3)) return MVIT_EUNSUPPORTED;
    (void)act_dtype;
    const int To = T / 2, So = S / 4;
    const int tiles = ((So + ST_TX - 1) / ST_TX) * ((So + ST_TY - 1) / ST_TY);
    dim3 grid(tiles, To, B);
    hipLaunchKernelGGL(stem_f32_kernel, grid, dim3(256), 0, as_stream(stream), clip, w, bias, pos_spatial, pos_temporal,
                       x, T, S, To, So);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}
