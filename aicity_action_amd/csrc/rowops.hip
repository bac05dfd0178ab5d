// Row-wise, HBM-bound kernels: LayerNorm, final LN + token-mean + head, skip max-pool, casts.
// All are bandwidth kernels: 16-byte coalesced accesses, no LDS except block reductions.
#include "common.h"

// ----------------------------------------------------------------------------------------------
// LayerNorm: x fp32 [rows][C] -> y TO [rows][C].  C/12 lanes per row, 3 float4 per lane.
// Two-pass (mean, then centred variance) in registers, like ATen's CPU kernel.
// ----------------------------------------------------------------------------------------------
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int off = LPR / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int C, typename TO>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, TO* __restrict__ y,
                                                     int64_t rows, float eps) {
    constexpr int LPR = C / 12;
    constexpr int RPB = 256 / LPR;
    const int lir = threadIdx.x % LPR;
    const int rib = threadIdx.x / LPR;
    float4 g[3], bt[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        g[i] = load4(gamma + 4 * (lir + LPR * i));
        bt[i] = load4(beta + 4 * (lir + LPR * i));
    }
    for (int64_t r0 = (int64_t)blockIdx.x * RPB; r0 < rows; r0 += (int64_t)gridDim.x * RPB) {
        const int64_t r = r0 + rib;
        const bool ok = r < rows;
        float4 v[3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
            v[i] = ok ? load4(x + r * C + 4 * (lir + LPR * i)) : make_float4(0.f, 0.f, 0.f, 0.f);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        const float mean = group_sum<LPR>(s) * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
            q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
        }
        const float rstd = 1.0f / sqrtf(group_sum<LPR>(q) * (1.0f / C) + eps);
        if (ok) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                float4 o;
                o.x = v[i].x * rstd * g[i].x + bt[i].x;
                o.y = v[i].y * rstd * g[i].y + bt[i].y;
                o.z = v[i].z * rstd * g[i].z + bt[i].z;
                o.w = v[i].w * rstd * g[i].w + bt[i].w;
                store4(y + r * C + 4 * (lir + LPR * i), o);
            }
        }
    }
}

template <int C>
static int launch_ln(const float* x, const float* g, const float* b, void* y, int64_t rows, float eps,
                     int act_dtype, hipStream_t st) {
    constexpr int RPB = 256 / (C / 12);
    int64_t blocks = (rows + RPB - 1) / RPB;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    if (act_dtype == MVIT_F32)
        hipLaunchKernelGGL((ln_fwd_kernel<C, float>), dim3((unsigned)blocks), dim3(256), 0, st, x, g, b, (float*)y, rows, eps);
    else
        hipLaunchKernelGGL((ln_fwd_kernel<C, bf16_t>), dim3((unsigned)blocks), dim3(256), 0, st, x, g, b, (bf16_t*)y, rows, eps);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}

extern "C" int mvit_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, int64_t rows,
                                  int C, float eps, int act_dtype, void* stream) {
    if (!x || !gamma || !beta || !y || rows < 0) return MVIT_EINVAL;
    if (act_dtype != MVIT_F32 && act_dtype != MVIT_BF16) return MVIT_EDTYPE;
    if (rows == 0) return MVIT_OK;
    hipStream_t st = as_stream(stream);
    switch (C) {
        case 96: return launch_ln<96>(x, gamma, beta, y, rows, eps, act_dtype, st);
        case 192: return launch_ln<192>(x, gamma, beta, y, rows, eps, act_dtype, st);
        case 384: return launch_ln<384>(x, gamma, beta, y, rows, eps, act_dtype, st);
        case 768: return launch_ln<768>(x, gamma, beta, y, rows, eps, act_dtype, st);
        default: return MVIT_EUNSUPPORTED;
    }
}

// ----------------------------------------------------------------------------------------------
// Head: final LN + token mean (deterministic two-stage reduction) + Linear + softmax.
// ----------------------------------------------------------------------------------------------
#define HEAD_CHUNK 32  // tokens per block of stage 1

template <int C>
__global__ __launch_bounds__(256) void head_ln_partial_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ part,
                                                              int N, int nchunks, float eps) {
    constexpr int LPR = C / 12;
    constexpr int RPB = 256 / LPR;
    __shared__ float red[RPB][C];
    const int b = blockIdx.y, ch = blockIdx.x;
    const int lir = threadIdx.x % LPR, rib = threadIdx.x / LPR;
    float4 g[3], bt[3], acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        g[i] = load4(gamma + 4 * (lir + LPR * i));
        bt[i] = load4(beta + 4 * (lir + LPR * i));
        acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int t0 = ch * HEAD_CHUNK;
    for (int tt = 0; tt < HEAD_CHUNK; tt += RPB) {
        const int t = t0 + tt + rib;
        const bool ok = (tt + rib) < HEAD_CHUNK && t < N;
        float4 v[3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
            v[i] = ok ? load4(x + ((int64_t)b * N + t) * C + 4 * (lir + LPR * i)) : make_float4(0.f, 0.f, 0.f, 0.f);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        const float mean = group_sum<LPR>(s) * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
            q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
        }
        const float rstd = 1.0f / sqrtf(group_sum<LPR>(q) * (1.0f / C) + eps);
        if (ok) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                acc[i].x += v[i].x * rstd * g[i].x + bt[i].x;
                acc[i].y += v[i].y * rstd * g[i].y + bt[i].y;
                acc[i].z += v[i].z * rstd * g[i].z + bt[i].z;
                acc[i].w += v[i].w * rstd * g[i].w + bt[i].w;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) *reinterpret_cast<float4*>(&red[rib][4 * (lir + LPR * i)]) = acc[i];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < RPB; ++r) s += red[r][c];
        part[((int64_t)b * nchunks + ch) * C + c] = s;
    }
}

__global__ __launch_bounds__(256) void head_project_kernel(const float* __restrict__ part, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ logits,
                                                           float* __restrict__ probs, int N, int nchunks, int C, int ncls) {
    extern __shared__ float sm[];  // z[C] + lg[ncls]
    float* z = sm;
    float* lg = sm + C;
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int k = 0; k < nchunks; ++k) s += part[((int64_t)b * nchunks + k) * C + c];
        z[c] = s / (float)N;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int j = wave; j < ncls; j += 4) {
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += z[c] * w[(int64_t)j * C + c];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) lg[j] = s + bias[j];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = -INFINITY;
        for (int j = 0; j < ncls; ++j) m = fmaxf(m, lg[j]);
        float den = 0.f;
        for (int j = 0; j < ncls; ++j) den += expf(lg[j] - m);
        for (int j = 0; j < ncls; ++j) {
            if (logits) logits[(int64_t)b * ncls + j] = lg[j];
            if (probs) probs[(int64_t)b * ncls + j] = expf(lg[j] - m) / den;
        }
    }
}

extern "C" int64_t mvit_head_workspace_bytes(int B, int N, int C) {
    int64_t nchunks = (N + HEAD_CHUNK - 1) / HEAD_CHUNK;
    return (int64_t)B * nchunks * C * (int64_t)sizeof(float);
}

// stage 1 only (training path): per-chunk sums of LN(x) over tokens -> workspace [B][nchunks][C], nchunks = ceil(N/32)
extern "C" int mvit_head_ln_partial(const float* x, const float* gamma, const float* beta, float* workspace, int B, int N,
                                    int C, float eps, void* stream) {
    if (!x || !gamma || !beta || !workspace || B <= 0 || N <= 0) return MVIT_EINVAL;
    hipStream_t st = as_stream(stream);
    const int nchunks = (N + HEAD_CHUNK - 1) / HEAD_CHUNK;
    dim3 grid(nchunks, B);
    switch (C) {
        case 96: hipLaunchKernelGGL((head_ln_partial_kernel<96>), grid, dim3(256), 0, st, x, gamma, beta, workspace, N, nchunks, eps); break;
        case 192: hipLaunchKernelGGL((head_ln_partial_kernel<192>), grid, dim3(256), 0, st, x, gamma, beta, workspace, N, nchunks, eps); break;
        case 384: hipLaunchKernelGGL((head_ln_partial_kernel<384>), grid, dim3(256), 0, st, x, gamma, beta, workspace, N, nchunks, eps); break;
        case 768: hipLaunchKernelGGL((head_ln_partial_kernel<768>), grid, dim3(256), 0, st, x, gamma, beta, workspace, N, nchunks, eps); break;
        default: return MVIT_EUNSUPPORTED;
    }
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}

extern "C" int mvit_head_fwd(const float* x, const float* gamma, const float* beta, const float* w_head,
                             const float* b_head, float* workspace, float* logits, float* probs, int B, int N, int C,
                             int num_classes, float eps, void* stream) {
    if (!x || !gamma || !beta || !w_head || !b_head || !workspace || B <= 0 || N <= 0 || num_classes <= 0)
        return MVIT_EINVAL;
    hipStream_t st = as_stream(stream);
    const int nchunks = (N + HEAD_CHUNK - 1) / HEAD_CHUNK;
    dim3 grid(nchunks, B);
    switch (C) {
        case 96: hipLaunchKernelGGL((head_ln_partial_kernel<96>), grid, dim3(256), 0, st, x, gamma, beta, workspace, N, nchunks, eps); break;
        case 192: hipLaunchKernelGGL((head_ln_partial_kernel<192>), grid, dim3(256), 0, st, x, gamma, beta, workspace, N, nchunks, eps); break;
        case 384: hipLaunchKernelGGL((head_ln_partial_kernel<384>), grid, dim3(256), 0, st, x, gamma, beta, workspace, N, nchunks, eps); break;
        case 768: hipLaunchKernelGGL((head_ln_partial_kernel<768>), grid, dim3(256), 0, st, x, gamma, beta, workspace, N, nchunks, eps); break;
        default: return MVIT_EUNSUPPORTED;
    }
    MVIT_LAUNCH_CHECK();
    hipLaunchKernelGGL(head_project_kernel, dim3(B), dim3(256), (C + num_classes) * sizeof(float), st, workspace, w_head,
                       b_head, logits, probs, N, nchunks, C, num_classes);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}

// ----------------------------------------------------------------------------------------------
// Skip-path max pool, k(1,3,3) s(1,2,2) p(0,1,1), channel-last fp32.
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_skip_kernel(const float* __restrict__ x, float* __restrict__ y, int BT,
                                                           int H, int W, int Ho, int Wo, int C4) {
    const int64_t total = (int64_t)BT * Ho * Wo * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        int64_t tok = i / C4;
        const int xo = (int)(tok % Wo); tok /= Wo;
        const int yo = (int)(tok % Ho);
        const int64_t bt = tok / Ho;
        // all nine window loads are issued unconditionally at clamped coordinates (taps outside the frame re-read an in-frame
        // element of the same window, which cannot change the maximum); with `continue` on the bounds tests every load sat
        // behind its own branch + vmcnt(0)
        float4 v[9];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            int yi = 2 * yo + dy - 1;
            yi = yi < 0 ? 0 : (yi >= H ? H - 1 : yi);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                int xi = 2 * xo + dx - 1;
                xi = xi < 0 ? 0 : (xi >= W ? W - 1 : xi);
                v[dy * 3 + dx] = load4(x + (((bt * H + yi) * W + xi) * C4 + c4) * 4);
            }
        }
        float4 m = v[0];
#pragma unroll
        for (int k = 1; k < 9; ++k) {
            m.x = fmaxf(m.x, v[k].x); m.y = fmaxf(m.y, v[k].y); m.z = fmaxf(m.z, v[k].z); m.w = fmaxf(m.w, v[k].w);
        }
        store4(y + i * 4, m);
    }
}

extern "C" int mvit_maxpool_skip_fwd(const float* x, float* y, int B, int T, int H, int W, int C, void* stream) {
    if (!x || !y || B <= 0 || T <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return MVIT_EINVAL;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int64_t total = (int64_t)B * T * Ho * Wo * (C / 4);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(maxpool_skip_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, y, B * T, H, W,
                       Ho, Wo, C / 4);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}

// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ s, bf16_t* __restrict__ d, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) d[i] = f32_to_bf16(s[i]);
}

extern "C" int mvit_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
    if (!src || !dst || n < 0) return MVIT_EINVAL;
    if (n == 0) return MVIT_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), src, (bf16_t*)dst, n);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}

// rows x cols fp32 -> 16-bit with an optional per-row-group factor (drop-path scale of the gradient that feeds the
// weight- and data-gradient GEMMs); cols % 8 == 0, 32 B in / 16 B out per thread.
__global__ __launch_bounds__(256) void cast_rows_kernel(const float* __restrict__ s, bf16_t* __restrict__ d, int64_t rows, int cols8,
                                                        const float* __restrict__ row_scale, int64_t rps) {
    const int64_t n8 = rows * cols8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const float sc = row_scale ? row_scale[(i / cols8) / rps] : 1.0f;
        float4 lo, hi;
        load8(s + 8 * i, lo, hi);
        uint4 o;
        o.x = pack_bf16x2(lo.x * sc, lo.y * sc); o.y = pack_bf16x2(lo.z * sc, lo.w * sc);
        o.z = pack_bf16x2(hi.x * sc, hi.y * sc); o.w = pack_bf16x2(hi.z * sc, hi.w * sc);
        *reinterpret_cast<uint4*>(d + 8 * i) = o;
    }
}

extern "C" int mvit_cast_rows_f32_to_bf16(const float* src, void* dst, int64_t rows, int cols, const float* row_scale,
                                          int64_t rows_per_scale, void* stream) {
    if (!src || !dst || rows < 0 || cols <= 0 || (row_scale && rows_per_scale <= 0)) return MVIT_EINVAL;
    if (cols % 8) return MVIT_EUNSUPPORTED;
    if (rows == 0) return MVIT_OK;
    int64_t blocks = (rows * (cols / 8) + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(cast_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), src, (bf16_t*)dst, rows, cols / 8,
                       row_scale, rows_per_scale);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}

// fp32 [R][C] -> 16-bit copy [R][C] (optional) and 16-bit TRANSPOSED copy [C][R] in one pass (GEMM weights after an
// optimizer step: the forward reads W, the data-gradient GEMM reads W^T).  64x64 tiles through LDS, both outputs coalesced.
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ s, bf16_t* __restrict__ d, bf16_t* __restrict__ dt,
                                                             int R, int C) {
    __shared__ bf16_t tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        bf16_t v = 0;
        if (r < R && c < C) {
            v = f32_to_bf16(s[(int64_t)r * C + c]);
            if (d) d[(int64_t)r * C + c] = v;
        }
        tile[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (r < R && c < C) dt[(int64_t)c * R + r] = tile[tx][i];
    }
}

extern "C" int mvit_cast_transpose_f32_to_bf16(const float* src, void* dst, void* dst_t, int rows, int cols, void* stream) {
    if (!src || !dst_t || rows <= 0 || cols <= 0) return MVIT_EINVAL;
    hipLaunchKernelGGL(cast_transpose_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, as_stream(stream), src, (bf16_t*)dst,
                       (bf16_t*)dst_t, rows, cols);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}

// Head split without pooling (blocks whose query has no pooling conv: MVIT.Q_POOL_ALL off, attention.py:14-15,239-246):
// out[b][g][n][d] = qkv[b][n][chan_off + g*96 + d]; the backward writes the gradient back into the fused slice.
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void head_split_kernel(T* __restrict__ qkv, int64_t ld, int chan_off, T* __restrict__ hs, int B, int heads,
                                                         int64_t N) {
    constexpr int CW = 16 / sizeof(T);          // elements per 16-byte access
    constexpr int CPR = 96 / CW;                // accesses per 96-channel head row
    const int64_t total = (int64_t)B * heads * N * CPR;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % CPR);
        int64_t rem = i / CPR;
        const int64_t n = rem % N; rem /= N;
        const int g = (int)(rem % heads);
        const int64_t b = rem / heads;
        T* pq = qkv + (b * N + n) * ld + chan_off + g * 96 + c * CW;
        T* ph = hs + i * CW;
        if (BWD) *reinterpret_cast<uint4*>(pq) = *reinterpret_cast<const uint4*>(ph);
        else *reinterpret_cast<uint4*>(ph) = *reinterpret_cast<const uint4*>(pq);
    }
}

template <bool BWD>
static int launch_head_split(void* qkv, int64_t ld, int chan_off, void* hs, int B, int heads, int64_t N, int act_dtype, void* stream) {
    if (!qkv || !hs || B <= 0 || heads <= 0 || N <= 0) return MVIT_EINVAL;
    if (act_dtype != MVIT_F32 && act_dtype != MVIT_BF16) return MVIT_EDTYPE;
    if ((ld & 7) || (chan_off & 7)) return MVIT_EUNSUPPORTED;
    const int64_t total = (int64_t)B * heads * N * (act_dtype == MVIT_F32 ? 24 : 12);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (act_dtype == MVIT_F32)
        hipLaunchKernelGGL((head_split_kernel<float, BWD>), dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (float*)qkv, ld, chan_off,
                           (float*)hs, B, heads, N);
    else
        hipLaunchKernelGGL((head_split_kernel<bf16_t, BWD>), dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (bf16_t*)qkv, ld,
                           chan_off, (bf16_t*)hs, B, heads, N);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}

extern "C" int mvit_head_split_fwd(const void* qkv, int64_t ld, int chan_off, void* out, int B, int heads, int64_t N, int act_dtype,
                                   void* stream) {
    return launch_head_split<false>(const_cast<void*>(qkv), ld, chan_off, out, B, heads, N, act_dtype, stream);
}

extern "C" int mvit_head_split_bwd(const void* dout, void* dqkv, int64_t ld, int chan_off, int B, int heads, int64_t N, int act_dtype,
                                   void* stream) {
    return launch_head_split<true>(dqkv, ld, chan_off, const_cast<void*>(dout), B, heads, N, act_dtype, stream);
}

// All GEMM weights of the model in ONE launch: desc[t] = {src, dst, dst_t, rows, cols, first_tile}; workgroup b finds its tensor by
// binary search over first_tile and handles one 64x64 tile of it (same tile body as cast_transpose_kernel).
struct CastDesc {
    const float* src;
    bf16_t* dst;
    bf16_t* dst_t;
    int rows, cols;
    int first_tile, pad;
};
__global__ __launch_bounds__(256) void cast_transpose_multi_kernel(const CastDesc* __restrict__ desc, int ntensors) {
    __shared__ bf16_t tile[64][66];
    int lo = 0, hi = ntensors - 1;
    const int b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (desc[mid].first_tile <= b) lo = mid; else hi = mid - 1;
    }
    const CastDesc d = desc[lo];
    const int tl = b - d.first_tile, tcx = (d.cols + 63) / 64;
    const int r0 = (tl / tcx) * 64, c0 = (tl % tcx) * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        bf16_t v = 0;
        if (r < d.rows && c < d.cols) {
            v = f32_to_bf16(d.src[(int64_t)r * d.cols + c]);
            d.dst[(int64_t)r * d.cols + c] = v;
        }
        tile[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (r < d.rows && c < d.cols) d.dst_t[(int64_t)c * d.rows + r] = tile[tx][i];
    }
}

extern "C" int mvit_cast_desc_bytes(void) { return (int)sizeof(CastDesc); }

extern "C" int mvit_cast_transpose_multi(const void* desc_table, int ntensors, int total_tiles, void* stream) {
    if (!desc_table || ntensors <= 0 || total_tiles <= 0) return MVIT_EINVAL;
    hipLaunchKernelGGL(cast_transpose_multi_kernel, dim3((unsigned)total_tiles), dim3(256), 0, as_stream(stream),
                       (const CastDesc*)desc_table, ntensors);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}

// Fork / join on the library's side stream for callers that issue independent operators themselves (the k / v pooling convs
// beside the q one): after mvit_side_fork(stream) the handle returned by mvit_side_stream() may be passed as the `stream`
// argument of any entry point; mvit_side_join(stream) orders everything issued there before later work on `stream`.
extern "C" void* mvit_side_stream(void) {
    SideStream* ss = side_stream_for_current_device();
    return ss ? reinterpret_cast<void*>(ss->side) : nullptr;
}
extern "C" int mvit_side_fork(void* stream) {
    SideStream* ss = side_stream_for_current_device();
    if (!ss) return MVIT_EUNSUPPORTED;
    return side_fork(ss, as_stream(stream)) ? MVIT_OK : MVIT_ELAUNCH;
}
extern "C" int mvit_side_join(void* stream) {
    SideStream* ss = side_stream_for_current_device();
    if (!ss) return MVIT_EUNSUPPORTED;
    return side_join(ss, as_stream(stream)) ? MVIT_OK : MVIT_ELAUNCH;
}

#ifdef MVIT_HALF_IS_FP16
extern "C" const char* mvit_version(void) { return "mvit-hip gfx950 r6 (16-bit type: fp16)"; }
#else
extern "C" const char* mvit_version(void) { return "mvit-hip gfx950 r6 (16-bit type: bf16)"; }
#endif

extern "C" const char* mvit_strerror(int code) {
    switch (code) {
        case MVIT_OK: return "ok";
        case MVIT_EINVAL: return "invalid argument (shape / null pointer)";
        case MVIT_EDTYPE: return "unsupported dtype";
        case MVIT_ELAUNCH: return "HIP launch error";
        case MVIT_EUNSUPPORTED: return "shape outside compiled specialisations";
        default: return "unknown error";
    }
}
