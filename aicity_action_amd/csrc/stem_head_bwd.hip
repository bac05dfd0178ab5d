// Backward of the two ends of the network.
//   stem : dW[96][3][3][7][7], dpos_spatial, dpos_temporal from d_x0 (bias grad = column sum, see mvit_colsum)
//   head : training forward variant that also returns the (dropout-masked) pooled feature z, and the tiny
//          backward through Linear(C, num_classes) + dropout + token mean (the final LayerNorm backward is
//          mvit_layernorm_bwd in broadcast mode).
#include "common.h"

// ------------------------------------------------------------------------------------------------
// stem weight gradient (v1, fp32 VALU): persistent workgroups walk 8x8-token tiles; thread = (channel, group of
// 10 taps) and keeps its 9 planes x 10 taps partial sums in registers; one fp32 atomic per (thread, tap) at the end.
// ------------------------------------------------------------------------------------------------
#define SB_PW 35
__global__ __launch_bounds__(512) void stem_wgrad_kernel(const float* __restrict__ clip, const float* __restrict__ dx,
                                                         float* __restrict__ dW, int B, int T, int S, int To, int So,
                                                         int tiles_x, int tiles_y) {
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
    if (on) {
#pragma unroll
        for (int p = 0; p < 9; ++p)
#pragma unroll
            for (int k = 0; k < 10; ++k)
                if (k < ntap) atomicAdd(dW + (int64_t)c * 441 + p * 49 + tap0 + k, acc[p][k]);
    }
}

// dpos_s[hw][c] = sum_{b,t} dx[b][t][hw][c] ;  dpos_t[t][c] = sum_{b,hw} dx[b][t][hw][c]  (the latter via fp32 atomics
// on T*96 outputs after a block-level partial sum)
__global__ __launch_bounds__(256) void stem_pos_bwd_kernel(const float* __restrict__ dx, float* __restrict__ dps,
                                                           float* __restrict__ dpt, int B, int To, int HW) {
    __shared__ float red[8][96];
    const int hw0 = blockIdx.x * 8;
    const int c4 = threadIdx.x % 24, r = threadIdx.x / 24;   // 10 row slots, 8 used
    const int t = blockIdx.y;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int hw = hw0 + r;
    const bool ok = r < 8 && hw < HW;
    if (ok)
        for (int b = 0; b < B; ++b) {
            const float4 v = load4(dx + (((int64_t)b * To + t) * HW + hw) * 96 + 4 * c4);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    if (ok) {   // spatial: one atomic per (hw,c) per t  (To atomics per output element)
        atomicAdd(dps + (int64_t)hw * 96 + 4 * c4, acc.x); atomicAdd(dps + (int64_t)hw * 96 + 4 * c4 + 1, acc.y);
        atomicAdd(dps + (int64_t)hw * 96 + 4 * c4 + 2, acc.z); atomicAdd(dps + (int64_t)hw * 96 + 4 * c4 + 3, acc.w);
    }
    if (r < 8) *reinterpret_cast<float4*>(&red[r][4 * c4]) = ok ? acc : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (threadIdx.x < 96) {
        float s = 0.f;
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) s += red[rr][threadIdx.x];
        atomicAdd(dpt + t * 96 + threadIdx.x, s);
    }
}

// dW, dpos_s, dpos_t are ACCUMULATED into (caller zeroes them once per step).
extern "C" int mvit_stem_bwd(const float* clip, const float* dx, float* dW, float* dpos_spatial, float* dpos_temporal,
                             int B, int T, int S, void* stream) {
    if (!clip || !dx || !dW || !dpos_spatial || !dpos_temporal || B <= 0 || T <= 0 || S <= 0) return MVIT_EINVAL;
    if ((T & 1) || (S & 3)) return MVIT_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
    const int To = T / 2, So = S / 4;
    const int tiles_x = (So + 7) / 8, tiles_y = (So + 7) / 8;
    const int ntiles = B * To * tiles_x * tiles_y;
    hipLaunchKernelGGL(stem_wgrad_kernel, dim3(ntiles < 512 ? ntiles : 512), dim3(512), 0, st, clip, dx, dW, B, T, S, To, So,
                       tiles_x, tiles_y);
    MVIT_LAUNCH_CHECK();
    dim3 grid((So * So + 7) / 8, To);
    hipLaunchKernelGGL(stem_pos_bwd_kernel, grid, dim3(256), 0, st, dx, dpos_spatial, dpos_temporal, B, To, So * So);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
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
