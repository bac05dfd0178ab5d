// Linear layers: y = epilogue(a . w^T).
//   * bf16 path: MFMA 32x32x16 bf16, 128x96 output tile per 256-thread workgroup, whole 96-wide
//     K slabs staged in LDS (every K of the model is a multiple of 96), fp32 accumulate, fused
//     bias / erf-GELU / drop-path scale / fp32 residual epilogue staged through LDS so global stores
//     are whole rows.
//   * fp32 path: exact-fp32 LDS-tiled VALU GEMM (parity path, not performance critical).
#include "common.h"

// ------------------------------------------------------------------------------------------------
// bf16 MFMA GEMM
// ------------------------------------------------------------------------------------------------
#define LBM 128
#define LBN 96
#define LBK 96
#define L_ROWB 192                    // bytes per LDS row (96 bf16)
#define L_STAGE_LD 100                // fp32 epilogue staging leading dim (floats)

// LDS image of a [rows][96] bf16 slab: 12 16-byte chunks per row, chunk c of row r stored at
// position (c + ((r>>2)&3)) % 12: ds_read_b128 by the 32x32x16 A/B fragment pattern (16-lane groups
// of rows {0-3,12-15,20-27} / {4-11,16-19,28-31}) then touches 16 distinct 16-B slots of the 256-B
// bank row -> conflict-free with unpadded 192-B rows.
__device__ __forceinline__ int slab_off(int row, int chunk) {
    int p = chunk + ((row >> 2) & 3);
    p = p >= 12 ? p - 12 : p;
    return row * L_ROWB + p * 16;
}

__device__ __forceinline__ uint4 load_chunk8(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ uint4 load_chunk8(const float* p) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    uint4 r;
    r.x = pack_bf16x2(a.x, a.y); r.y = pack_bf16x2(a.z, a.w);
    r.z = pack_bf16x2(b.x, b.y); r.w = pack_bf16x2(b.z, b.w);
    return r;
}

// bijective XCD-aware remap: workgroups that share blockIdx%8 (one XCD under round-robin dispatch)
// get a contiguous range of logical tiles, so the n-tiles of one A panel hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    const int base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

template <typename TA, typename TO>
__global__ __launch_bounds__(256) void linear_mfma_kernel(
    const TA* __restrict__ a, int64_t lda, const bf16_t* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ residual, int64_t ldr, const float* __restrict__ row_scale, int64_t rows_per_scale,
    TO* __restrict__ y, int64_t ldy, int64_t M, int N, int K, int epilogue) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;
    char* sB = smem + LBM * L_ROWB;
    float* stage = reinterpret_cast<float*>(smem);

    const int ntn = N / LBN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = tile % ntn;
    const int64_t tm = tile / ntn;
    const int64_t m0 = tm * LBM;
    const int n0 = tn * LBN;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;

    // staging assignment: chunk q = tid + 256*i -> row q/12, chunk q%12
    int a_row[6], a_chk[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int q = tid + 256 * i;
        a_row[i] = q / 12;
        a_chk[i] = q - a_row[i] * 12;
    }
    int b_row[5], b_chk[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int q = tid + 256 * i;
        b_row[i] = q / 12;
        b_chk[i] = q - b_row[i] * 12;
    }

    f32x16 acc[3];
#pragma unroll
    for (int nb = 0; nb < 3; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;

    // per-lane fragment offsets (same for A rows 32*wave+r and B rows 32*nb+r: (row>>2)&3 == (r>>2)&3)
    int foff[6];
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
        int p = 2 * ks + h + ((r >> 2) & 3);
        p = p >= 12 ? p - 12 : p;
        foff[ks] = p * 16;
    }
    const char* fa = sA + (32 * wave + r) * L_ROWB;
    const char* fb = sB + r * L_ROWB;

    uint4 ra[6], rb[5];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            int64_t m = m0 + a_row[i];
            m = m < M ? m : M - 1;  // clamp: rows >= M are never stored
            ra[i] = load_chunk8(a + m * lda + k0 + 8 * a_chk[i]);
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            if (i < 4 || tid < 128) rb[i] = load_chunk8(w + (int64_t)(n0 + b_row[i]) * K + k0 + 8 * b_chk[i]);
        }
    };

    const int nk = K / LBK;
    gload(0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();  // previous slab's fragment reads are done
#pragma unroll
        for (int i = 0; i < 6; ++i) *reinterpret_cast<uint4*>(sA + slab_off(a_row[i], a_chk[i])) = ra[i];
#pragma unroll
        for (int i = 0; i < 5; ++i)
            if (i < 4 || tid < 128) *reinterpret_cast<uint4*>(sB + slab_off(b_row[i], b_chk[i])) = rb[i];
        __syncthreads();
        if (kt + 1 < nk) gload((kt + 1) * LBK);  // prefetch next slab under the MFMAs
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(fa + foff[ks]);
#pragma unroll
            for (int nb = 0; nb < 3; ++nb) {
                const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(fb + nb * 32 * L_ROWB + foff[ks]);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[nb], 0, 0, 0);
            }
        }
    }
    __syncthreads();  // slabs dead -> reuse LDS as the fp32 staging tile

    // ---- epilogue part 1: bias / GELU in accumulator layout, stage to LDS -----------------------
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) {
        const int n = nb * 32 + r;
        const float bv = (epilogue & MVIT_EPI_BIAS) ? bias[n0 + n] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = 32 * wave + (i & 3) + 8 * (i >> 2) + 4 * h;
            float v = acc[nb][i] + bv;
            if (epilogue & MVIT_EPI_GELU) v = gelu_erf(v);
            stage[m * L_STAGE_LD + n] = v;
        }
    }
    __syncthreads();
    // ---- epilogue part 2: whole-row coalesced pass: scale, residual, store ------------------------
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const int q = tid + 256 * i;        // 128 rows x 24 float4
        const int row = q / 24, c4 = q - row * 24;
        const int64_t m = m0 + row;
        if (m < M) {
            float4 v = *reinterpret_cast<const float4*>(stage + row * L_STAGE_LD + 4 * c4);
            if (row_scale) {
                const float s = row_scale[m / rows_per_scale];
                v.x *= s; v.y *= s; v.z *= s; v.w *= s;
            }
            if (epilogue & MVIT_EPI_RESIDUAL) {
                const float4 rr = load4(residual + m * ldr + n0 + 4 * c4);
                v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
            }
            store4(y + m * ldy + n0 + 4 * c4, v);
        }
    }
}

#define L_SMEM_BYTES (LBM * L_STAGE_LD * 4)  // 51200 >= slabs (43008)

template <typename TA, typename TO>
static int launch_linear_mfma(const void* a, int64_t lda, const void* w, const float* bias, const float* residual,
                              int64_t ldr, const float* row_scale, int64_t rps, void* y, int64_t ldy, int64_t M, int N,
                              int K, int epi, hipStream_t st) {
    const int64_t ntm = (M + LBM - 1) / LBM;
    const int64_t nwg = ntm * (N / LBN);
    if (nwg > 0x7fffffff) return MVIT_EINVAL;
    hipLaunchKernelGGL((linear_mfma_kernel<TA, TO>), dim3((unsigned)nwg), dim3(256), L_SMEM_BYTES, st, (const TA*)a, lda,
                       (const bf16_t*)w, bias, residual, ldr, row_scale, rps, (TO*)y, ldy, M, N, K, epi);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}

// ------------------------------------------------------------------------------------------------
// exact fp32 GEMM (VALU), 64x64 tile, BK 16, 4x4 per thread
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void linear_f32_kernel(const float* __restrict__ a, int64_t lda,
                                                         const float* __restrict__ w, const float* __restrict__ bias,
                                                         const float* __restrict__ residual, int64_t ldr,
                                                         const float* __restrict__ row_scale, int64_t rows_per_scale,
                                                         float* __restrict__ y, int64_t ldy, int64_t M, int N, int K,
                                                         int epilogue) {
    __shared__ float As[16][68];
    __shared__ float Ws[16][68];
    const int tid = threadIdx.x;
    const int ntn = (N + 63) / 64;
    const int tn = blockIdx.x % ntn;
    const int64_t tm = blockIdx.x / ntn;
    const int64_t m0 = tm * 64;
    const int n0 = tn * 64;
    const int tx = tid & 15, ty = tid >> 4;  // thread computes rows ty*4.., cols tx*4..
    const int lr = tid >> 2, lk = (tid & 3) * 4;  // loader: row lr (0..63), k offset lk
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
        {
            int64_t m = m0 + lr; m = m < M ? m : M - 1;
            const float4 v = load4(a + m * lda + k0 + lk);
            As[lk + 0][lr] = v.x; As[lk + 1][lr] = v.y; As[lk + 2][lr] = v.z; As[lk + 3][lr] = v.w;
            int n = n0 + lr; n = n < N ? n : N - 1;
            const float4 u = load4(w + (int64_t)n * K + k0 + lk);
            Ws[lk + 0][lr] = u.x; Ws[lk + 1][lr] = u.y; Ws[lk + 2][lr] = u.z; Ws[lk + 3][lr] = u.w;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const float4 av = *reinterpret_cast<const float4*>(&As[kk][ty * 4]);
            const float4 wv = *reinterpret_cast<const float4*>(&Ws[kk][tx * 4]);
            const float ar[4] = {av.x, av.y, av.z, av.w};
            const float wr[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(ar[i], wr[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + ty * 4 + i;
        if (m >= M) continue;
        const float s = row_scale ? row_scale[m / rows_per_scale] : 1.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= N) continue;
            float v = acc[i][j];
            if (epilogue & MVIT_EPI_BIAS) v += bias[n];
            if (epilogue & MVIT_EPI_GELU) v = gelu_erf(v);
            v *= s;
            if (epilogue & MVIT_EPI_RESIDUAL) v += residual[m * ldr + n];
            y[m * ldy + n] = v;
        }
    }
}

extern "C" int mvit_linear_fwd(const void* a, int a_dtype, int64_t lda, const void* w, const float* bias,
                               const float* residual, int64_t ldr, const float* row_scale, int64_t rows_per_scale,
                               void* y, int out_dtype, int64_t ldy, int64_t M, int N, int K, int epilogue, int act_dtype,
                               void* stream) {
    if (!a || !w || !y || M < 0 || N <= 0 || K <= 0) return MVIT_EINVAL;
    if ((epilogue & MVIT_EPI_BIAS) && !bias) return MVIT_EINVAL;
    if ((epilogue & MVIT_EPI_RESIDUAL) && !residual) return MVIT_EINVAL;
    if (row_scale && rows_per_scale <= 0) return MVIT_EINVAL;
    if (M == 0) return MVIT_OK;
    hipStream_t st = as_stream(stream);
    if (act_dtype == MVIT_F32) {
        if (a_dtype != MVIT_F32 || out_dtype != MVIT_F32) return MVIT_EDTYPE;
        if ((K & 15) || (lda & 3)) return MVIT_EUNSUPPORTED;
        const int64_t nwg = ((M + 63) / 64) * ((N + 63) / 64);
        if (nwg > 0x7fffffff) return MVIT_EINVAL;
        hipLaunchKernelGGL(linear_f32_kernel, dim3((unsigned)nwg), dim3(256), 0, st, (const float*)a, lda, (const float*)w,
                           bias, residual, ldr, row_scale, rows_per_scale, (float*)y, ldy, M, N, K, epilogue);
        MVIT_LAUNCH_CHECK();
        return MVIT_OK;
    }
    if (act_dtype != MVIT_BF16) return MVIT_EDTYPE;
    if (K % LBK || N % LBN || (lda & 7) || (ldy & 3) || ((epilogue & MVIT_EPI_RESIDUAL) && (ldr & 3)))
        return MVIT_EUNSUPPORTED;
#define DISPATCH(TA, TO) \
    return launch_linear_mfma<TA, TO>(a, lda, w, bias, residual, ldr, row_scale, rows_per_scale, y, ldy, M, N, K, epilogue, st)
    if (a_dtype == MVIT_BF16 && out_dtype == MVIT_BF16) DISPATCH(bf16_t, bf16_t);
    if (a_dtype == MVIT_BF16 && out_dtype == MVIT_F32) DISPATCH(bf16_t, float);
    if (a_dtype == MVIT_F32 && out_dtype == MVIT_F32) DISPATCH(float, float);
    if (a_dtype == MVIT_F32 && out_dtype == MVIT_BF16) DISPATCH(float, bf16_t);
#undef DISPATCH
    return MVIT_EDTYPE;
}
