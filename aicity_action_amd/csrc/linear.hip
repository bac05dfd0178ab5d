// Linear layers: y = epilogue(a . w^T).
//   * bf16 path: MFMA 32x32x16 bf16, 128x96 output tile per 256-thread workgroup, whole 96-wide
//     K slabs staged in LDS (every K of the model is a multiple of 96), fp32 accumulate, fused
//     bias / erf-GELU / drop-path scale / fp32 residual epilogue staged through LDS so global stores
//     are whole rows.
//   * fp32 path: exact-fp32 LDS-tiled VALU GEMM (parity path, not performance critical).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

// linear_pp.hip: the 256 x 192 ping-pong kernel (epilogue codes PP_*)
bool mvit_internal_linear_pp_ok(int64_t lda, int64_t M, int N, int K);
int mvit_internal_linear_pp(int epi, const void* a, int64_t lda, const void* w, const float* bias, const void* aux, int64_t ldaux,
                            const float* row_scale, int64_t rps, void* y, void* y2, int64_t ldy, int64_t M, int N, int K, hipStream_t st);
enum { PP_B16 = 0, PP_GELU16 = 1, PP_GELU_PRE = 2, PP_GELU_DER = 3, PP_F32 = 4, PP_F32_RES = 5, PP_F32_RES_SC = 6, PP_DG_PRE = 7, PP_DG_DER = 8 };
// linear_k96.hip: K = 96 layers on long token streams, weights resident in LDS (epilogue codes 0..3 = PP_B16 .. PP_GELU_DER)
bool mvit_internal_linear_k96_ok(int64_t lda, int64_t M, int N, int K);
int mvit_internal_linear_k96(int epi, const void* a, int64_t lda, const void* w, const float* bias, void* y, void* y2, int64_t ldy, int64_t M,
                             int N, hipStream_t st);
// Which shapes go to it (MVIT_GEMM_PP=0 / 1 forces off / on for every shape it can take): measured per shape in profiles/r3_gemm_shapes.txt
static inline bool use_pp(int64_t lda, int64_t M, int N, int K, int code) {
    static const char* env = getenv("MVIT_GEMM_PP");
    if (env && env[0] == '0') return false;
    if (!mvit_internal_linear_pp_ok(lda, M, N, K)) return false;
    if (env && env[0] == '1') return true;
    // Measured INSIDE the model, kernel by kernel (profiles/r3_gemm_in_model_ab.txt): in isolation the ping-pong kernel is ahead on every
    // shape (profiles/r3_gemm_pp_final_ab.txt), inside a step only where the epilogue is short -- 16-bit bias outputs (qkv and the
    // long-K data gradients: -9 % over a train step), the K <= 512 fp32-residual output (proj) and the two-output fc1 of training.
    // The GELU and long-K fp32-residual epilogues run under the OTHER workgroup's main loop in the 128 x 192 kernels (two workgroups
    // per CU) and stay there.
    if (M < 8192) return false;
    if (code == PP_DG_PRE || code == PP_DG_DER || code == PP_GELU16) return false;
    if ((code == PP_F32_RES || code == PP_F32_RES_SC) && (N < 384 || K > 512)) return false;
    return true;
}

// ------------------------------------------------------------------------------------------------
// bf16 MFMA GEMM
// ------------------------------------------------------------------------------------------------
#define LBM 128
#define LBN 96
#define LBK 96
#define L_ROWB 192                    // bytes per LDS row (96 bf16)
#define L_STAGE_LD 100                // fp32 epilogue staging leading dim (floats)
#define L_SMEM_BYTES (LBM * L_STAGE_LD * 4)  // 51200 >= slabs of the 128-row tile (43008)

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

// Staging map: 16 lanes per row (12 carry a 16-byte chunk, 4 idle), rows tid/16 + 16*i.  The rotation
// ((row>>2)&3) is then the same for every i, so LDS offsets are base + i*16*192 (immediates) and global
// offsets are base + i*16*lda: no per-slab index arithmetic.
template <typename TA, typename TO, int WM>   // WM = 32-row m-blocks per wave: tile = (128*WM) x 96
__global__ __launch_bounds__(256, 2) void linear_mfma_kernel(
    const TA* __restrict__ a, int64_t lda, const bf16_t* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ residual, int64_t ldr, const float* __restrict__ row_scale, int64_t rows_per_scale,
    TO* __restrict__ y, int64_t ldy, int64_t M, int N, int K, int epilogue) {
    constexpr int BM = LBM * WM;
    constexpr int NA = BM / 16;                // A row groups per thread per slab (8 or 16)
    constexpr int NB = LBN / 16;               // 6
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;
    char* sB = smem + BM * L_ROWB;
    float* stage = reinterpret_cast<float*>(smem);

    const int ntn = N / LBN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = tile % ntn;
    const int64_t tm = tile / ntn;
    const int64_t m0 = tm * BM;
    const int n0 = tn * LBN;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;

    const int srow = tid >> 4, schk = tid & 15;
    const bool s_on = schk < 12;
    const int s_lds = slab_off(srow, s_on ? schk : 0);   // + i*16*L_ROWB
    const bool full_m = m0 + BM <= M;
    const TA* a_ptr = a + (m0 + srow) * lda + 8 * (s_on ? schk : 0);
    const bf16_t* w_ptr = w + (int64_t)(n0 + srow) * K + 8 * (s_on ? schk : 0);

    f32x16 acc[WM][3];
#pragma unroll
    for (int mb = 0; mb < WM; ++mb)
#pragma unroll
        for (int nb = 0; nb < 3; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;

    // per-lane fragment offsets (same for A rows 32*j+r and B rows 32*nb+r: (row>>2)&3 == (r>>2)&3)
    int foff[6];
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
        int p = 2 * ks + h + ((r >> 2) & 3);
        p = p >= 12 ? p - 12 : p;
        foff[ks] = p * 16;
    }
    const char* fa = sA + (32 * WM * wave + r) * L_ROWB;
    const char* fb = sB + r * L_ROWB;

    uint4 ra[NA], rb[NB];
    auto gload = [&](int k0) {
        if (s_on) {
            if (full_m) {
#pragma unroll
                for (int i = 0; i < NA; ++i) ra[i] = load_chunk8(a_ptr + (int64_t)i * 16 * lda + k0);
            } else {
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    int64_t m = m0 + srow + 16 * i;
                    m = m < M ? m : M - 1;  // clamp: rows >= M are never stored
                    ra[i] = load_chunk8(a + m * lda + 8 * schk + k0);
                }
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) rb[i] = load_chunk8(w_ptr + (int64_t)i * 16 * K + k0);
        }
    };

    const int nk = K / LBK;
    gload(0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();  // previous slab's fragment reads are done
        if (s_on) {
#pragma unroll
            for (int i = 0; i < NA; ++i) *reinterpret_cast<uint4*>(sA + s_lds + i * 16 * L_ROWB) = ra[i];
#pragma unroll
            for (int i = 0; i < NB; ++i) *reinterpret_cast<uint4*>(sB + s_lds + i * 16 * L_ROWB) = rb[i];
        }
        __syncthreads();
        if (kt + 1 < nk) gload((kt + 1) * LBK);  // prefetch next slab under the MFMAs
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            bf16x8 af[WM];
#pragma unroll
            for (int mb = 0; mb < WM; ++mb) af[mb] = *reinterpret_cast<const bf16x8*>(fa + mb * 32 * L_ROWB + foff[ks]);
#pragma unroll
            for (int nb = 0; nb < 3; ++nb) {
                const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(fb + nb * 32 * L_ROWB + foff[ks]);
#pragma unroll
                for (int mb = 0; mb < WM; ++mb)
                    acc[mb][nb] = mfma16(af[mb], bfr, acc[mb][nb]);
            }
        }
    }

    // ---- epilogue, one 128-row pass per m-block: bias/GELU in accumulator layout -> fp32 LDS stage ->
    //      whole-row coalesced pass (drop-path scale, residual, 16-byte stores) ------------------------------
    constexpr int CWO = 16 / sizeof(TO);       // output elements per 16-byte store
    constexpr int CPR = LBN / CWO;             // chunks per row (12 bf16 / 24 fp32)
    constexpr int LPR = (CPR == 12) ? 16 : 32; // lanes per row (CPR of them active)
    constexpr int RPI = 256 / LPR;             // rows per iteration
    const int erow = tid / LPR, ec = tid % LPR;
    const bool e_on = ec < CPR;
#pragma unroll
    for (int mb = 0; mb < WM; ++mb) {
        __syncthreads();  // slabs (or the previous pass's stage) are dead
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) {
            const int n = nb * 32 + r;
            const float bv = (epilogue & MVIT_EPI_BIAS) ? bias[n0 + n] : 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ml = 32 * wave + (i & 3) + 8 * (i >> 2) + 4 * h;
                float v = acc[mb][nb][i] + bv;
                if (epilogue & MVIT_EPI_GELU) v = gelu_fast(v);
                stage[ml * L_STAGE_LD + n] = v;
            }
        }
        __syncthreads();
        // rows -> global: flags decided once, residual loads of all of this thread's rows in flight before the first add
        auto emit = [&](auto res_tag, auto scale_tag, auto full_tag) {
            constexpr bool RES = decltype(res_tag)::value, SCL = decltype(scale_tag)::value, FULL = decltype(full_tag)::value;
            constexpr int NR = 128 / RPI;
            float4 rr[RES ? NR : 1][RES ? CWO / 4 : 1];
            if (RES) {
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    const int row = erow + RPI * i;
                    int64_t m = m0 + (row >> 5) * (32 * WM) + 32 * mb + (row & 31);
                    m = (FULL || m < M) ? m : M - 1;
#pragma unroll
                    for (int e = 0; e < CWO; e += 4) rr[RES ? i : 0][RES ? e / 4 : 0] = load4(residual + m * ldr + n0 + CWO * ec + e);
                }
            }
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int row = erow + RPI * i;
                // stage row (32*wave' + l) holds tile row 32*WM*wave' + 32*mb + l
                const int64_t m = m0 + (row >> 5) * (32 * WM) + 32 * mb + (row & 31);
                const bool ok = FULL || m < M;
                float v[CWO];
#pragma unroll
                for (int e = 0; e < CWO; e += 4) {
                    const float4 t4 = *reinterpret_cast<const float4*>(stage + row * L_STAGE_LD + CWO * ec + e);
                    v[e] = t4.x; v[e + 1] = t4.y; v[e + 2] = t4.z; v[e + 3] = t4.w;
                }
                if (SCL) {
                    const float sc = row_scale[(ok ? m : M - 1) / rows_per_scale];
#pragma unroll
                    for (int e = 0; e < CWO; ++e) v[e] *= sc;
                }
                if (RES) {
#pragma unroll
                    for (int e = 0; e < CWO; e += 4) {
                        const float4 r4 = rr[RES ? i : 0][RES ? e / 4 : 0];
                        v[e] += r4.x; v[e + 1] += r4.y; v[e + 2] += r4.z; v[e + 3] += r4.w;
                    }
                }
                if (ok) {
                    TO* dst = y + m * ldy + n0 + CWO * ec;
                    if constexpr (sizeof(TO) == 2) {
                        uint4 o;
                        o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
                        o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
                        *reinterpret_cast<uint4*>(dst) = o;
                    } else {
                        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                }
            }
        };
        if (e_on) {
            const bool res = (epilogue & MVIT_EPI_RESIDUAL) != 0, scl = row_scale != nullptr;
            using T_ = std::true_type;
            using F_ = std::false_type;
#define EMIT3(R, S) { if (full_m) emit(R{}, S{}, T_{}); else emit(R{}, S{}, F_{}); }
            if (res && scl) EMIT3(T_, T_)
            else if (res) EMIT3(T_, F_)
            else if (scl) EMIT3(F_, T_)
            else EMIT3(F_, F_)
#undef EMIT3
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant (bf16 A): slabs are 48 k wide (96-byte rows), double-buffered, filled by
// global_load_lds_dwordx4 (no VGPR staging, no ds_write), one barrier per slab.  The DMA writes LDS
// linearly in lane order, so the bank-conflict swizzle is applied on the per-lane SOURCE address:
// LDS chunk position P = row*6 + p holds logical chunk c = (p - ((row>>4)&1)) mod 6 of that row, and the
// fragment reader applies the same rotation (ds_read_b128 by the 32x32x16 pattern is then conflict-free).
// ------------------------------------------------------------------------------------------------
#define DK 48
#define D_ROWB 96
#define D_A_BYTES (LBM * D_ROWB)   // 12288
#define D_B_BYTES (LBN * D_ROWB)   // 9216
#define D_BUF (D_A_BYTES + D_B_BYTES)

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

template <typename TO>
__global__ __launch_bounds__(256, 2) void linear_dma_kernel(
    const bf16_t* __restrict__ a, int64_t lda, const bf16_t* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ residual, int64_t ldr, const float* __restrict__ row_scale, int64_t rows_per_scale,
    TO* __restrict__ y, int64_t ldy, int64_t M, int N, int K, int epilogue) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* stage = reinterpret_cast<float*>(smem);

    const int ntn = N / LBN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = tile % ntn;
    const int64_t tm = tile / ntn;
    const int64_t m0 = tm * LBM;
    const int n0 = tn * LBN;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const bool full_m = m0 + LBM <= M;

    // DMA source offsets (elements, relative to the tile's first row), fixed for the whole K loop
    int a_off[3], b_off[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int P = 64 * (3 * wave + i) + lane;          // A: 768 chunks, 12 wave-instructions
        int row = P / 6;
        const int ph = P - row * 6;
        int c = ph - ((row >> 4) & 1);
        c = c < 0 ? c + 6 : c;
        if (!full_m && m0 + row >= M) row = (int)(M - 1 - m0);
        a_off[i] = row * (int)lda + 8 * c;
        const int Pb = 64 * (4 * i + wave) + lane;         // B: 576 chunks, 9 wave-instructions (wave 0 issues 3)
        int rowb = Pb / 6;
        const int phb = Pb - rowb * 6;
        int cb = phb - ((rowb >> 4) & 1);
        cb = cb < 0 ? cb + 6 : cb;
        rowb = rowb < LBN ? rowb : LBN - 1;
        b_off[i] = rowb * K + 8 * cb;
    }
    const bf16_t* a_tile = a + m0 * lda;
    const bf16_t* w_tile = w + (int64_t)n0 * K;
    const int nb_instr = (wave == 0) ? 3 : 2;

    auto dma = [&](int k0, int buf) {
        char* base = smem + buf * D_BUF;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t*)(a_tile + a_off[i] + k0), (lptr_t*)(base + 1024 * (3 * wave + i)), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < nb_instr)
                __builtin_amdgcn_global_load_lds((gptr_t*)(w_tile + b_off[i] + k0),
                                                 (lptr_t*)(base + D_A_BYTES + 1024 * (4 * i + wave)), 16, 0, 0);
    };

    f32x16 acc[3];
#pragma unroll
    for (int nb = 0; nb < 3; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
    int foff[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
        int p = 2 * ks + h + ((r >> 4) & 1);
        p = p >= 6 ? p - 6 : p;
        foff[ks] = p * 16;
    }
    const int fa_off = (32 * wave + r) * D_ROWB;
    const int fb_off = D_A_BYTES + r * D_ROWB;

    const int nk = K / DK;
    dma(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's DMA pieces of slab kt have landed
        __syncthreads();                      // ... and everyone's; all reads of the other buffer are done
        if (kt + 1 < nk) dma((kt + 1) * DK, (kt + 1) & 1);
        const char* base = smem + (kt & 1) * D_BUF;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(base + fa_off + foff[ks]);
#pragma unroll
            for (int nb = 0; nb < 3; ++nb) {
                const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(base + fb_off + nb * 32 * D_ROWB + foff[ks]);
                acc[nb] = mfma16(af, bfr, acc[nb]);
            }
        }
    }

    constexpr int CWO = 16 / sizeof(TO);
    constexpr int CPR = LBN / CWO;
    constexpr int LPR = (CPR == 12) ? 16 : 32;
    constexpr int RPI = 256 / LPR;
    const int erow = tid / LPR, ec = tid % LPR;
    const bool e_on = ec < CPR;
    __syncthreads();  // slabs dead
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) {
        const int n = nb * 32 + r;
        const float bv = (epilogue & MVIT_EPI_BIAS) ? bias[n0 + n] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ml = 32 * wave + (i & 3) + 8 * (i >> 2) + 4 * h;
            float v = acc[nb][i] + bv;
            if (epilogue & MVIT_EPI_GELU) v = gelu_fast(v);
            stage[ml * L_STAGE_LD + n] = v;
        }
    }
    __syncthreads();
    // rows -> global: the run-time flags are decided once; the residual loads of all of this thread's rows are issued before the
    // first add (inside the unrolled loop the flag tests made every load wait vmcnt(0) on its own)
    auto emit = [&](auto res_tag, auto scale_tag, auto full_tag) {
        constexpr bool RES = decltype(res_tag)::value, SCL = decltype(scale_tag)::value, FULL = decltype(full_tag)::value;
        constexpr int NR = 128 / RPI;
        float4 rr[RES ? NR : 1][RES ? CWO / 4 : 1];
        if (RES) {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                int64_t m = m0 + erow + RPI * i;
                m = (FULL || m < M) ? m : M - 1;
#pragma unroll
                for (int e = 0; e < CWO; e += 4) rr[RES ? i : 0][RES ? e / 4 : 0] = load4(residual + m * ldr + n0 + CWO * ec + e);
            }
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int row = erow + RPI * i;
            const int64_t m = m0 + row;
            const bool ok = FULL || m < M;
            float v[CWO];
#pragma unroll
            for (int e = 0; e < CWO; e += 4) {
                const float4 t4 = *reinterpret_cast<const float4*>(stage + row * L_STAGE_LD + CWO * ec + e);
                v[e] = t4.x; v[e + 1] = t4.y; v[e + 2] = t4.z; v[e + 3] = t4.w;
            }
            if (SCL) {
                const float sc = row_scale[(ok ? m : M - 1) / rows_per_scale];
#pragma unroll
                for (int e = 0; e < CWO; ++e) v[e] *= sc;
            }
            if (RES) {
#pragma unroll
                for (int e = 0; e < CWO; e += 4) {
                    const float4 r4 = rr[RES ? i : 0][RES ? e / 4 : 0];
                    v[e] += r4.x; v[e + 1] += r4.y; v[e + 2] += r4.z; v[e + 3] += r4.w;
                }
            }
            if (ok) {
                TO* dst = y + m * ldy + n0 + CWO * ec;
                if constexpr (sizeof(TO) == 2) {
                    uint4 o;
                    o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
                    o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
                    *reinterpret_cast<uint4*>(dst) = o;
                } else {
                    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    };
    if (e_on) {
        const bool res = (epilogue & MVIT_EPI_RESIDUAL) != 0, scl = row_scale != nullptr;
        using T_ = std::true_type;
        using F_ = std::false_type;
#define EMIT3(R, S) { if (full_m) emit(R{}, S{}, T_{}); else emit(R{}, S{}, F_{}); }
        if (res && scl) EMIT3(T_, T_)
        else if (res) EMIT3(T_, F_)
        else if (scl) EMIT3(F_, T_)
        else EMIT3(F_, F_)
#undef EMIT3
    }
}

template <typename TO>
static int launch_linear_dma(const void* a, int64_t lda, const void* w, const float* bias, const float* residual,
                             int64_t ldr, const float* row_scale, int64_t rps, void* y, int64_t ldy, int64_t M, int N,
                             int K, int epi, hipStream_t st) {
    const int64_t nwg = ((M + LBM - 1) / LBM) * (N / LBN);
    if (nwg > 0x7fffffff) return MVIT_EINVAL;
    hipLaunchKernelGGL((linear_dma_kernel<TO>), dim3((unsigned)nwg), dim3(256), L_SMEM_BYTES, st, (const bf16_t*)a, lda,
                       (const bf16_t*)w, bias, residual, ldr, row_scale, rps, (TO*)y, ldy, M, N, K, epi);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}

// ------------------------------------------------------------------------------------------------
// Large-tile LDS-DMA variant: 128 x 192 output tile, 4 waves (2 along M x 2 along N, 64x96 each),
// 64-wide K slabs = one whole 128-byte line per row, double-buffered in LDS (80 KB -> 2 workgroups/CU) and
// filled by global_load_lds_dwordx4 one slab ahead (raw s_barrier; the DMA stays in flight under the MFMAs).
//
// Measured reasons for this shape (rocprof PMC + ablation builds, M=50176 N=1152 K=384, MI355X):
//   * 96-byte K segments straddle 128-B lines: ~1 GB of L2->L1 line traffic for 0.45 GB useful, 50 % of wave
//     time in s_waitcnt  -> whole-line slabs, bigger tile.
//   * an fp32 LDS-staged epilogue cost 40 % of the kernel at 1 workgroup/CU  -> the product is computed
//     TRANSPOSED (A operand = weight rows, B operand = token rows), so each lane owns one output ROW and
//     4 consecutive columns per accumulator quad; bias/GELU/residual are applied in registers and rows are
//     stored straight from registers in 16-byte pieces (bf16: quads of the two half-waves are first paired
//     with v_permlane32_swap).  No LDS round trip; 2 workgroups/CU overlap one's epilogue with the other's MFMAs.
// LDS image: 128-byte rows, chunk c of row r at position c ^ ((r>>1)&7) (conflict-free ds_read_b128 for the
// 32x32x16 fragment pattern); the DMA writes linearly, so the XOR is applied to the per-lane SOURCE address.
// ------------------------------------------------------------------------------------------------
#define G_BM 128
#define G_BN 192
#define G_BK 64
#define G_ROWB 128
#define G_PANEL_A (G_BM * G_ROWB)     // 16384
#define G_PANEL_B (G_BN * G_ROWB)     // 24576
#define G_BUF (G_PANEL_A + G_PANEL_B) // 40960
#define G_SMEM (2 * G_BUF)            // 81920
#define G_KOK(K) ((K) % G_BK == 0 || ((K) % G_BK == 32 && (K) >= G_BK))      // K the 128x192 kernels cover (see ktail)

// LDS fragment reads of the persistent kernel are inline asm: the compiler cannot tell a ds_read from the LDS-DMA
// writes still in flight for the next slab and (depending on how it peels the loop) puts s_waitcnt vmcnt(0) in front of
// the first MFMA, which serialises the DMA ring; it also waits lgkmcnt(0) after every other read.  Here all 20 reads
// of a slab are issued first and each k-step waits only for its own five (LDS returns in order).
template <int OFF>
__device__ __forceinline__ bf16x8 lds_read128(uint32_t addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N>
__device__ __forceinline__ void lds_wait5(bf16x8& a, bf16x8& b, bf16x8& c, bf16x8& d, bf16x8& e) {
    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : "n"(N));
}

#ifndef BIG_ABL
#define BIG_ABL 0      // timing ablations of linear_big_kernel (tools/r6_big_ab.sh; results invalid): 1 no epilogue stores, 2 no aux loads, 4 aux loads NOT prefetched (round-5 form),
                       // 8 stores as contiguous 1-KiB runs (wrong places, same bytes), 16 aux loads as contiguous 1-KiB runs (wrong data, same bytes),
                       // 32 one workgroup of a CU in its main loop at a time (per-CU token; results VALID),
                       // 64 the second resident workgroup of every CU starts BIG_DELAY x 64 x 127 cycles late (one-time symmetry breaking; results VALID),
                       // 128 the first round's workgroups of XCD k start k x BIG_DELAY x 2560 cycles late (results VALID),
                       // 256 s_memrealtime stamps (100 MHz, chip-wide) at workgroup start / main-loop end / kernel end + the CU id, read back by mvit_debug_big_stamps (results VALID)
#endif
// DG (16-bit out only): `residual` carries the 16-bit pre-activation of the MLP (row stride ldr) and the result is multiplied by
// GELU'(pre): the data gradient of fc2 leaves the GEMM as the gradient of fc1's output (no separate element-wise pass)
#if BIG_ABL & 256
__device__ unsigned long long g_big_stamps[8192 * 8];
extern "C" __attribute__((visibility("default"))) int mvit_debug_big_stamps(unsigned long long* out) {      /* tools/r6_big_stamps.py only; not part of the C-ABI */
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_big_stamps), sizeof(unsigned long long) * 8192 * 8) == hipSuccess ? 0 : -3;
}
#endif
#if BIG_ABL & 32
// Probe: a per-CU token around the main loop.  The two workgroups of a CU share the matrix pipes, which pulls them into lock-step (the
// one ahead slows down whenever both multiply, the one behind speeds up whenever the other stores): main loops together, then epilogues
// together, memory idle in the first phase and the matrix cores in the second.  With the token only one of them multiplies at a time.
__device__ unsigned g_cu_token[16 * 256];
__device__ __forceinline__ unsigned cu_key() {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);      // XCC_ID [3:0]
    return (xcc & 15u) * 256u + ((hw >> 8) & 255u);
}
#endif
template <typename TO, bool RES, bool SCALE, int DG = 0>     // DG: 1 = aux is the pre-activation, 2 = aux is GELU'(pre) already
__global__ __launch_bounds__(256, 2) void linear_big_kernel(
    const bf16_t* __restrict__ a, int64_t lda, const bf16_t* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ residual, int64_t ldr, const float* __restrict__ row_scale, int64_t rows_per_scale,
    TO* __restrict__ y, int64_t ldy, int64_t M, int N, int K, int epilogue) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int ntn = N / G_BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = tile % ntn;
    const int64_t tm = tile / ntn;
    const int64_t m0 = tm * G_BM;
    const int n0 = tn * G_BN;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
#if BIG_ABL & 256
    const unsigned long long st_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    const bool full_m = m0 + G_BM <= M;

    // DMA piece p (1 KiB = 8 rows x 128 B): lane -> row 8p + lane/8, position lane%8, logical chunk pos ^ swz(row)
    // A panel: 16 pieces (4 per wave); B panel: 24 pieces (6 per wave)
    int a_off[4], b_off[6];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int row = 8 * (4 * wave + i) + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        if (!full_m && m0 + row >= M) row = (int)(M - 1 - m0);
        a_off[i] = row * (int)lda + 8 * c;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int row = 8 * (6 * wave + i) + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        b_off[i] = row * K + 8 * c;
    }
    const bf16_t* a_tile = a + m0 * lda;
    const bf16_t* w_tile = w + (int64_t)n0 * K;

    auto dma = [&](int k0, int buf) {
        char* base = smem + buf * G_BUF;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t*)(a_tile + a_off[i] + k0), (lptr_t*)(base + 1024 * (4 * wave + i)), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 6; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t*)(w_tile + b_off[i] + k0),
                                             (lptr_t*)(base + G_PANEL_A + 1024 * (6 * wave + i)), 16, 0, 0);
    };

    // acc[mb][nb]: rows (registers) = n = 96*wn + 32*nb + (i&3) + 8*(i>>2) + 4*h ; column (lane) = m = 64*wm + 32*mb + r
    f32x16 acc[2][3];
    if (sizeof(TO) == 4 && (epilogue & MVIT_EPI_BIAS)) {
        // fp32 output: the bias is the accumulator's initial value (wave-uniform scalar loads, no vector load in the epilogue)
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) {
            const float4* bp = reinterpret_cast<const float4*>(bias + n0 + 96 * wn + 32 * nb);
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
    } else {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    uint32_t lds_x[4], lds_w[4];       // per-lane LDS byte addresses of the k-step fragments in buffer 0
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int fo = ((2 * ks + h) ^ ((r >> 1) & 7)) * 16;
        lds_x[ks] = lds0 + (64 * wm + r) * G_ROWB + fo;                 // token rows  (MFMA B operand)
        lds_w[ks] = lds0 + G_PANEL_A + (96 * wn + r) * G_ROWB + fo;     // weight rows (MFMA A operand)
    }

    // K = 64 j + 32 (the 96-wide layers of block 0): the last slab is loaded from column K - 64, so nothing is read past a row,
    // and its first two k-steps -- columns the previous slab already covered -- are multiplied with zeroed token fragments
    const int nk = (K + G_BK - 1) / G_BK;
    const bool ktail = (K & (G_BK - 1)) != 0;
    // DG: the epilogue multiplies every output by a 16-bit operand of the output's shape (the saved GELU' / the pre-activation).  Round 6:
    // its twelve 16-byte loads per lane are REQUESTED UNDER THE MAIN LOOP (behind the DMA of slab 1, so that slab 1's wait may leave
    // them outstanding: vmcnt retires in order) instead of after the last MFMA, where both workgroups of a CU sat out the full memory
    // latency together before their first store (profiles/r6_gemm_big_prefetch_ab.txt).  48 registers; the kernel has 256.
    constexpr bool PREFETCH = DG != 0 && !(BIG_ABL & 4) && !(BIG_ABL & 2);
    [[maybe_unused]] uint4 pre_all[PREFETCH ? 2 : 1][3][2];
    // fp32 output with a residual: the residual row piece of the FIRST 32-row block (twelve 16-byte loads, 48 registers) is requested the
    // same way; the second block's is requested at the top of the epilogue, in front of the first block's arithmetic and stores
    constexpr bool PREFETCH_R = sizeof(TO) == 4 && RES && !(BIG_ABL & 4) && !(BIG_ABL & 2);
    [[maybe_unused]] float4 rr_pre[PREFETCH_R ? 3 : 1][PREFETCH_R ? 4 : 1];
#if BIG_ABL & 64
#ifndef BIG_DELAY
#define BIG_DELAY 2
#endif
    // first round of workgroups: ids 0..255 land in slot 0 of the 256 CUs, 256..511 in slot 1 (8 XCDs x 32 CUs, round-robin) -- the second slot starts late
    if (blockIdx.x >= 256 && blockIdx.x < 512) {
#pragma unroll 1
        for (int i = 0; i < BIG_DELAY; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif
#if BIG_ABL & 128
#ifndef BIG_DELAY
#define BIG_DELAY 2
#endif
    // probe: the first round's workgroups of XCD k (= blockIdx.x % 8) start k x BIG_DELAY x 2560 cycles late: the eight XCDs' store bursts no longer coincide
    if (blockIdx.x < 512) {
#pragma unroll 1
        for (int i = 0; i < (int)(blockIdx.x & 7) * BIG_DELAY; ++i) __builtin_amdgcn_s_sleep(40);
    }
#endif
#ifndef BIG_PIPE
#define BIG_PIPE 1      // slabs in flight: 1 = the product's loop; 2 = the two-slab probe of round 6 (below; -DBIG_PIPE=2)
#endif
#if BIG_PIPE == 2
    // PROBE, not the product (profiles/r6_gemm_big_pipe_ab.txt: fc2 data gradient +3-5 %, long-K residual -2-4 %, train step unchanged).
    // TWO slabs in flight on TWO LDS buffers (round 6).  The stamps of profiles/r6_gemm_big_stamps.txt put a K-tile at 2.0 us of which the 24 MFMAs are 0.36 us:
    // with one slab of lead (requested at the top of iteration kt, awaited at the top of kt + 1) an iteration cannot be shorter than the LDS-DMA's issue-to-landed
    // time (~1.1 us, MI355X_MICROARCH.md) and a CU's two workgroups keep ~40 KB in flight, 40 GB/s.  Here a wave reads ALL 20 fragments of slab kt into registers
    // first (it holds them all anyway), a second barrier says the buffer is free, and slab kt + 2 is requested into it BEFORE the MFMAs of slab kt: a request has
    // an iteration and a half to land, two slabs per workgroup are in flight.  Same slabs, same k order per accumulator: results bit-identical.
    auto k0_of = [&](int s_) { return (ktail && s_ + 1 == nk) ? K - G_BK : s_ * G_BK; };
    dma(0, 0);
    if (nk > 1) dma(k0_of(1), 1);
    for (int kt = 0; kt < nk; ++kt) {
        // this wave's pieces of slab kt have landed; younger and allowed to fly: slab kt + 1 (10 pieces) and, in iterations 1 and 2, the 12 epilogue-operand
        // loads requested in iteration 0 (vmcnt retires in order; they sit between slab 2 and slab 3 in the issue order)
        {
            const int young = ((kt + 1 < nk) ? 10 : 0) + (((PREFETCH || PREFETCH_R) && (kt == 1 || kt == 2)) ? 12 : 0);
            if (young == 22) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
            else if (young == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (young == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                      // everyone's pieces of slab kt have landed
        const bool half = ktail && kt + 1 == nk;
        const uint32_t bo = (kt & 1) ? G_BUF : 0;
        bf16x8 xf[4][2], wf[4][3];
#define RD(KS) { const uint32_t xa = lds_x[KS] + bo, wa = lds_w[KS] + bo; \
                 xf[KS][0] = lds_read128<0>(xa); xf[KS][1] = lds_read128<32 * G_ROWB>(xa); \
                 wf[KS][0] = lds_read128<0>(wa); wf[KS][1] = lds_read128<32 * G_ROWB>(wa); wf[KS][2] = lds_read128<64 * G_ROWB>(wa); }
#define MM(KS) _Pragma("unroll") for (int nb = 0; nb < 3; ++nb) _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) \
                 acc[mb][nb] = mfma16(wf[KS][nb], xf[KS][mb], acc[mb][nb]);
        RD(0) RD(1) RD(2) RD(3)
        lds_wait5<0>(xf[0][0], xf[0][1], wf[0][0], wf[0][1], wf[0][2]);
        lds_wait5<0>(xf[1][0], xf[1][1], wf[1][0], wf[1][1], wf[1][2]);
        lds_wait5<0>(xf[2][0], xf[2][1], wf[2][0], wf[2][1], wf[2][2]);
        lds_wait5<0>(xf[3][0], xf[3][1], wf[3][0], wf[3][1], wf[3][2]);
        __builtin_amdgcn_s_barrier();                      // every wave holds its fragments of slab kt: buffer kt & 1 is free
        if (kt + 2 < nk) dma(k0_of(kt + 2), kt & 1);
        if constexpr (PREFETCH) {
            if (kt == 0) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    int64_t m = m0 + 64 * wm + 32 * mb + r;
                    m = (full_m || m < M) ? m : M - 1;
                    const bf16_t* pre = reinterpret_cast<const bf16_t*>(residual) + m * ldr + n0 + 96 * wn;
#pragma unroll
                    for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                        for (int qq = 0; qq < 2; ++qq) {
                            if constexpr (BIG_ABL & 16) {
                                const int run = ((wave * 2 + mb) * 3 + nb) * 2 + qq, piece = run * 64 + lane;
                                const int prow = piece / 24, pc = piece - prow * 24;
                                pre_all[mb][nb][qq] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(residual) + (m0 + prow) * ldr + n0 + 8 * pc);
                            } else
                            pre_all[mb][nb][qq] = *reinterpret_cast<const uint4*>(pre + 32 * nb + 8 * (2 * qq + h));
                        }
                }
            }
        }
        if constexpr (PREFETCH_R) {
            if (kt == 0) {
                int64_t m = m0 + 64 * wm + r;
                m = (full_m || m < M) ? m : M - 1;
#pragma unroll
                for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) rr_pre[nb][q] = load4(residual + m * ldr + n0 + 96 * wn + 32 * nb + 4 * h + 8 * q);
            }
        }
        if (half) { xf[0][0] = bf16x8{}; xf[0][1] = bf16x8{}; xf[1][0] = bf16x8{}; xf[1][1] = bf16x8{}; }
        MM(0)
        __builtin_amdgcn_sched_barrier(0);
        MM(1)
        __builtin_amdgcn_sched_barrier(0);
        MM(2)
        __builtin_amdgcn_sched_barrier(0);
        MM(3)
#undef RD
#undef MM
    }
#else
    dma(0, 0);
#if BIG_ABL & 32
    const unsigned ckey = cu_key();
    if (tid == 0) {
        while (atomicCAS(&g_cu_token[ckey], 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(4);
    }
#endif
#if BIG_ABL & 256
    unsigned long long ph_wait = 0, ph_bar = 0, ph_dma = 0, ph_mm = 0;
#define PH_T(v) const unsigned long long v = __builtin_readcyclecounter();
#else
#define PH_T(v)
#endif
    for (int kt = 0; kt < nk; ++kt) {
        PH_T(p0_)
        if ((PREFETCH || PREFETCH_R) && kt == 1 && nk > 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");      // slab 1 has landed; the 12 younger loads may still fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of slab kt have landed
        PH_T(p1_)
        __builtin_amdgcn_s_barrier();                      // everyone's have; everyone is done reading the other buffer
        PH_T(p2_)
        if (kt + 1 < nk) dma((ktail && kt + 2 == nk) ? K - G_BK : (kt + 1) * G_BK, (kt + 1) & 1);
        PH_T(p3_)
        if constexpr (PREFETCH) {
            if (kt == 0) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    int64_t m = m0 + 64 * wm + 32 * mb + r;
                    m = (full_m || m < M) ? m : M - 1;
                    const bf16_t* pre = reinterpret_cast<const bf16_t*>(residual) + m * ldr + n0 + 96 * wn;
#pragma unroll
                    for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                        for (int qq = 0; qq < 2; ++qq) {
                            if constexpr (BIG_ABL & 16) {
                                const int run = ((wave * 2 + mb) * 3 + nb) * 2 + qq, piece = run * 64 + lane;
                                const int prow = piece / 24, pc = piece - prow * 24;
                                pre_all[mb][nb][qq] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(residual) + (m0 + prow) * ldr + n0 + 8 * pc);
                            } else
                            pre_all[mb][nb][qq] = *reinterpret_cast<const uint4*>(pre + 32 * nb + 8 * (2 * qq + h));
                        }
                }
            }
        }
        if constexpr (PREFETCH_R) {
            if (kt == 0) {
                int64_t m = m0 + 64 * wm + r;
                m = (full_m || m < M) ? m : M - 1;
#pragma unroll
                for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) rr_pre[nb][q] = load4(residual + m * ldr + n0 + 96 * wn + 32 * nb + 4 * h + 8 * q);
            }
        }
        const bool half = ktail && kt + 1 == nk;
        const uint32_t bo = (kt & 1) ? G_BUF : 0;
        bf16x8 xf[4][2], wf[4][3];
#define RD(KS) { const uint32_t xa = lds_x[KS] + bo, wa = lds_w[KS] + bo; \
                 xf[KS][0] = lds_read128<0>(xa); xf[KS][1] = lds_read128<32 * G_ROWB>(xa); \
                 wf[KS][0] = lds_read128<0>(wa); wf[KS][1] = lds_read128<32 * G_ROWB>(wa); wf[KS][2] = lds_read128<64 * G_ROWB>(wa); }
#define MM(KS) _Pragma("unroll") for (int nb = 0; nb < 3; ++nb) _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) \
                 acc[mb][nb] = mfma16(wf[KS][nb], xf[KS][mb], acc[mb][nb]);
        RD(0) RD(1)
        lds_wait5<5>(xf[0][0], xf[0][1], wf[0][0], wf[0][1], wf[0][2]);
        if (half) { xf[0][0] = bf16x8{}; xf[0][1] = bf16x8{}; }
        RD(2)
        MM(0)
        __builtin_amdgcn_sched_barrier(0);
        lds_wait5<5>(xf[1][0], xf[1][1], wf[1][0], wf[1][1], wf[1][2]);
        if (half) { xf[1][0] = bf16x8{}; xf[1][1] = bf16x8{}; }
        RD(3)
        MM(1)
        __builtin_amdgcn_sched_barrier(0);
        lds_wait5<5>(xf[2][0], xf[2][1], wf[2][0], wf[2][1], wf[2][2]);
        MM(2)
        __builtin_amdgcn_sched_barrier(0);
        lds_wait5<0>(xf[3][0], xf[3][1], wf[3][0], wf[3][1], wf[3][2]);
        MM(3)
#undef RD
#undef MM
#if BIG_ABL & 256
        { const unsigned long long p4_ = __builtin_readcyclecounter(); ph_wait += p1_ - p0_; ph_bar += p2_ - p1_; ph_dma += p3_ - p2_; ph_mm += p4_ - p3_; }
#endif
    }

#endif      // BIG_PIPE
#if BIG_ABL & 32
    __builtin_amdgcn_s_barrier();
    if (tid == 0) atomicExch(&g_cu_token[ckey], 0u);
#endif
#if BIG_ABL & 256
    const unsigned long long st_t1 = __builtin_amdgcn_s_memrealtime();
    struct StampAtExit {
        unsigned long long t0, t1; unsigned bid; int tid; unsigned long long pw, pb, pd, pm;
        __device__ ~StampAtExit() {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the epilogue's stores have been acknowledged
            if (tid == 0 && bid < 8192) {
                const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
                g_big_stamps[8 * bid] = t0; g_big_stamps[8 * bid + 1] = t1; g_big_stamps[8 * bid + 2] = __builtin_amdgcn_s_memrealtime();
                g_big_stamps[8 * bid + 3] = (unsigned long long)((xcc & 15u) * 256u + ((hw >> 8) & 255u));
                g_big_stamps[8 * bid + 4] = pw; g_big_stamps[8 * bid + 5] = pb; g_big_stamps[8 * bid + 6] = pd; g_big_stamps[8 * bid + 7] = pm;
            }
        }
    } stamp_at_exit{st_t0, st_t1, blockIdx.x, tid,
#if BIG_PIPE == 1
                    ph_wait, ph_bar, ph_dma, ph_mm
#else
                    0, 0, 0, 0
#endif
    };
#endif
    if (epilogue & 256) {   // DIAG build aid: skip the epilogue, keep the accumulators live
        float t = 0.f;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                for (int i = 0; i < 16; ++i) t += acc[mb][nb][i];
        if (t == 12345.678f) y[0] = (TO)0;
        return;
    }
    if constexpr (sizeof(TO) == 4) {
        // fp32 output: y = residual + scale * gelu?(acc) with the bias already inside acc.  Straight-line code: the twelve 16-byte
        // residual loads of a 32-row block are all in flight before the first add (the runtime-flag version waited vmcnt(0)
        // after every single load); the GELU flag and the ragged-M mask are decided once, outside the unrolled loops.
        auto emit = [&](auto full_tag, auto gelu_tag) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                int64_t m = m0 + 64 * wm + 32 * mb + r;
                const bool ok = decltype(full_tag)::value || m < M;
                m = ok ? m : M - 1;
                float sc = 1.f;
                if (SCALE) sc = row_scale[m / rows_per_scale];
                float4 rr[3][4];
                if (RES) {
                    if (PREFETCH_R && mb == 0) {           // block 0: requested under the main loop; block 1: request it now, use it after block 0's stores
#pragma unroll
                        for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                            for (int q = 0; q < 4; ++q) rr[nb][q] = rr_pre[PREFETCH_R ? nb : 0][PREFETCH_R ? q : 0];
                        int64_t m1 = m0 + 64 * wm + 32 + r;
                        m1 = (decltype(full_tag)::value || m1 < M) ? m1 : M - 1;
#pragma unroll
                        for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                            for (int q = 0; q < 4; ++q) rr_pre[PREFETCH_R ? nb : 0][PREFETCH_R ? q : 0] = load4(residual + m1 * ldr + n0 + 96 * wn + 32 * nb + 4 * h + 8 * q);
                    } else if (PREFETCH_R) {
#pragma unroll
                        for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                            for (int q = 0; q < 4; ++q) rr[nb][q] = rr_pre[PREFETCH_R ? nb : 0][PREFETCH_R ? q : 0];
                    } else if (BIG_ABL & 2) {
#pragma unroll
                        for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                            for (int q = 0; q < 4; ++q) rr[nb][q] = make_float4(1.f, 1.f, 1.f, 1.f);
                    } else {
#pragma unroll
                        for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                            for (int q = 0; q < 4; ++q) rr[nb][q] = load4(residual + m * ldr + n0 + 96 * wn + 32 * nb + 4 * h + 8 * q);
                    }
                }
#pragma unroll
                for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float4 v = make_float4(acc[mb][nb][4 * q], acc[mb][nb][4 * q + 1], acc[mb][nb][4 * q + 2], acc[mb][nb][4 * q + 3]);
                        if (decltype(gelu_tag)::value) { v.x = gelu_fast(v.x); v.y = gelu_fast(v.y); v.z = gelu_fast(v.z); v.w = gelu_fast(v.w); }
                        if (SCALE) { v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc; }
                        if (RES) { v.x += rr[nb][q].x; v.y += rr[nb][q].y; v.z += rr[nb][q].z; v.w += rr[nb][q].w; }
                        if ((BIG_ABL & 1) ? (ok && v.x == 12345.678f) : ok) *reinterpret_cast<float4*>(y + m * ldy + n0 + 96 * wn + 32 * nb + 4 * h + 8 * q) = v;
                    }
            }
        };
        if (epilogue & MVIT_EPI_GELU) {
            if (full_m) emit(std::true_type{}, std::true_type{}); else emit(std::false_type{}, std::true_type{});
        } else {
            if (full_m) emit(std::true_type{}, std::false_type{}); else emit(std::false_type{}, std::false_type{});
        }
        return;
    }
    // ---- 16-bit output, epilogue from registers: lane = output row, quads of 4 consecutive columns -----------
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int64_t m = m0 + 64 * wm + 32 * mb + r;
        const bool ok = full_m || m < M;
        const float sc = (row_scale && ok) ? row_scale[m / rows_per_scale] : 1.f;
        [[maybe_unused]] uint4 pre_raw[3][2];
        if constexpr (DG) {     // the six 16-byte pieces of this row's pre-activation, in the OUTPUT piece layout
            if constexpr (PREFETCH) {
#pragma unroll
                for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) pre_raw[nb][qq] = pre_all[mb][nb][qq];
            } else if constexpr (BIG_ABL & 2) {
#pragma unroll
                for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) pre_raw[nb][qq] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
            } else {            // (round-5 form: all requested here, after the main loop)
                const bf16_t* pre = reinterpret_cast<const bf16_t*>(residual) + (ok ? m : M - 1) * ldr + n0 + 96 * wn;
#pragma unroll
                for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) pre_raw[nb][qq] = *reinterpret_cast<const uint4*>(pre + 32 * nb + 8 * (2 * qq + h));
            }
        }
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) {
            const int nbase = n0 + 96 * wn + 32 * nb + 4 * h;    // + 8*q
            float4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v[q] = make_float4(acc[mb][nb][4 * q], acc[mb][nb][4 * q + 1], acc[mb][nb][4 * q + 2], acc[mb][nb][4 * q + 3]);
                if (epilogue & MVIT_EPI_BIAS) {
                    const float4 bb = load4(bias + nbase + 8 * q);
                    v[q].x += bb.x; v[q].y += bb.y; v[q].z += bb.z; v[q].w += bb.w;
                }
                if (epilogue & MVIT_EPI_GELU) {
                    v[q].x = gelu_fast(v[q].x); v[q].y = gelu_fast(v[q].y);
                    v[q].z = gelu_fast(v[q].z); v[q].w = gelu_fast(v[q].w);
                }
                if (row_scale) { v[q].x *= sc; v[q].y *= sc; v[q].z *= sc; v[q].w *= sc; }
            }
            if constexpr (DG) {
                // a piece holds columns 8q'..8q'+7 (q' = 2qq+h): the inverse of the store-side exchange hands every lane the
                // pre-activations of its own quads q = 2qq and 2qq+1 (columns 8q + 4h .. +3)
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    const uint4 L = pre_raw[nb][qq];
                    const auto t0 = __builtin_amdgcn_permlane32_swap(L.x, L.z, false, false);
                    const auto t1 = __builtin_amdgcn_permlane32_swap(L.y, L.w, false, false);
                    float4& va = v[2 * qq];
                    float4& vb = v[2 * qq + 1];
                    if constexpr (DG == 2) {
                        va.x *= lo16_to_f32(t0[0]); va.y *= hi16_to_f32(t0[0]); va.z *= lo16_to_f32(t1[0]); va.w *= hi16_to_f32(t1[0]);
                        vb.x *= lo16_to_f32(t0[1]); vb.y *= hi16_to_f32(t0[1]); vb.z *= lo16_to_f32(t1[1]); vb.w *= hi16_to_f32(t1[1]);
                    } else {
                        va.x *= gelu_grad_fast(lo16_to_f32(t0[0])); va.y *= gelu_grad_fast(hi16_to_f32(t0[0]));
                        va.z *= gelu_grad_fast(lo16_to_f32(t1[0])); va.w *= gelu_grad_fast(hi16_to_f32(t1[0]));
                        vb.x *= gelu_grad_fast(lo16_to_f32(t0[1])); vb.y *= gelu_grad_fast(hi16_to_f32(t0[1]));
                        vb.z *= gelu_grad_fast(lo16_to_f32(t1[1])); vb.w *= gelu_grad_fast(hi16_to_f32(t1[1]));
                    }
                }
            }
            if constexpr (sizeof(TO) == 4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (ok) {
                        if (epilogue & MVIT_EPI_RESIDUAL) {
                            const float4 rr = load4(residual + m * ldr + nbase + 8 * q);
                            v[q].x += rr.x; v[q].y += rr.y; v[q].z += rr.z; v[q].w += rr.w;
                        }
                        *reinterpret_cast<float4*>(y + m * ldy + nbase + 8 * q) = v[q];
                    }
                }
            } else {
                // bf16 (no residual): lane (r,0) holds cols 8q..8q+3, lane (r,1) holds 8q+4..8q+7 of the same row.
                // permlane32_swap(a = quad q, b = quad q+1): afterwards the lower half holds 8 consecutive columns of
                // quad-pair q (16 B), the upper half those of quad-pair q+1.
#pragma unroll
                for (int q = 0; q < 4; q += 2) {
                    uint32_t a0 = pack_bf16x2(v[q].x, v[q].y), a1 = pack_bf16x2(v[q].z, v[q].w);
                    uint32_t b0 = pack_bf16x2(v[q + 1].x, v[q + 1].y), b1 = pack_bf16x2(v[q + 1].z, v[q + 1].w);
                    const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                    // lower half: {own q lo4 | upper's q (cols +4)} ; upper half: {lower's q+1 | own q+1 (cols +4)}
                    const uint4 o = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                    if constexpr (BIG_ABL & 8) {
                        // the tile's 48 KiB as 48 runs of 1 KiB: run = (wave, mb, nb, q/2), lane-linear inside; rows of the tile are ldy apart, a tile row holds 384 B
                        const int run = ((wave * 2 + mb) * 3 + nb) * 2 + (q >> 1), piece = run * 64 + lane;      // 16-byte piece index 0..3071 inside the tile
                        const int prow = piece / 24, pc = piece - prow * 24;
                        *reinterpret_cast<uint4*>(y + (m0 + prow) * ldy + n0 + 8 * pc) = o;
                    } else
                    if ((BIG_ABL & 1) ? (o.x == 0x12345678u && ok) : ok) *reinterpret_cast<uint4*>(y + m * ldy + (n0 + 96 * wn + 32 * nb) + 8 * (q + h)) = o;
                }
            }
        }
    }
}

template <typename TO, bool RES, bool SCALE, int DG = 0>
static int launch_linear_big_t(const void* a, int64_t lda, const void* w, const float* bias, const float* residual,
                               int64_t ldr, const float* row_scale, int64_t rps, void* y, int64_t ldy, int64_t M, int N,
                               int K, int epi, hipStream_t st) {
    const int64_t nwg = ((M + G_BM - 1) / G_BM) * (N / G_BN);
    if (nwg > 0x7fffffff) return MVIT_EINVAL;
    static DevFlags attr_done_tab; DevFlag attr_done = dev_flag(attr_done_tab);
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_big_kernel<TO, RES, SCALE, DG>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, G_SMEM) != hipSuccess)
            return MVIT_ELAUNCH;
        attr_done = true;
    }
    hipLaunchKernelGGL((linear_big_kernel<TO, RES, SCALE, DG>), dim3((unsigned)nwg), dim3(256), G_SMEM, st, (const bf16_t*)a, lda,
                       (const bf16_t*)w, bias, residual, ldr, row_scale, rps, (TO*)y, ldy, M, N, K, epi);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}

template <typename TO>
static int launch_linear_big(const void* a, int64_t lda, const void* w, const float* bias, const float* residual,
                             int64_t ldr, const float* row_scale, int64_t rps, void* y, int64_t ldy, int64_t M, int N,
                             int K, int epi, hipStream_t st) {
#define BIGL(R, S) return launch_linear_big_t<TO, R, S>(a, lda, w, bias, residual, ldr, row_scale, rps, y, ldy, M, N, K, epi, st)
    if constexpr (sizeof(TO) == 4) {
        const bool res = (epi & MVIT_EPI_RESIDUAL) != 0, scl = row_scale != nullptr;
        if (res && scl) BIGL(true, true);
        if (res) BIGL(true, false);
        if (scl) BIGL(false, true);
        BIGL(false, false);
    } else {
        BIGL(false, false);         // 16-bit output keeps the run-time flags (rare: only with a drop-path scale on a 16-bit result)
    }
#undef BIGL
}

// ------------------------------------------------------------------------------------------------
// Persistent form of the 128x192 kernel.  Each workgroup walks a strided list of tiles inside its XCD's
// contiguous tile range and treats (tile, slab) as ONE stream: while the last slab of a tile is multiplied the
// first slab of the next tile is already in flight, so the output stores of a tile overlap the next tile's
// load latency instead of ending the workgroup.  The bias is the accumulator's initial value (scalar loads at
// tile start), which leaves the common epilogues (bias, bias+GELU) free of vector loads: the stores trail and the
// next slab wait uses vmcnt(#stores) instead of vmcnt(0).
// ------------------------------------------------------------------------------------------------
// GELU: 0 = bias only, 1 = bias + GELU, 2 = both (16-bit out): y = GELU(pre) and y2 = pre, the pair a training step keeps;
// 3 = y = GELU(pre) and y2 = GELU'(pre): the backward then only multiplies (no transcendental in its epilogue)
// WM = waves along M: 2 = the 128 x 192 tile (4 waves, two workgroups per CU); 4 = a 256 x 192 tile (8 waves, 112 KiB of LDS, one
// workgroup per CU: 7 instead of 10 LDS-DMA pieces per wave and slab, the weight panel re-read from L2 half as often)
template <typename TO, int GELU, int WM = 2>   // no residual / row scale (those use linear_big_kernel)
__global__ __launch_bounds__(128 * WM, WM == 2 ? 2 : 1) void linear_pers_kernel(
    const bf16_t* __restrict__ a, int64_t lda, const bf16_t* __restrict__ w, const float* __restrict__ bias,
    TO* __restrict__ y2, int64_t ldr, const float* __restrict__ row_scale, int64_t rows_per_scale,
    TO* __restrict__ y, int64_t ldy, int64_t M, int N, int K, int epilogue) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NST = (sizeof(TO) == 4 ? 24 : 12) * (GELU >= 2 ? 2 : 1);     // vector stores per wave per full tile
    constexpr int BM = 64 * WM, PANEL_A = BM * G_ROWB, BUF = PANEL_A + G_PANEL_B, NB = 12 / WM;   // NB: weight-panel pieces per wave

    const int ntn = N / G_BN;
    const int nt = (int)((M + BM - 1) / BM) * ntn;
    const int per = gridDim.x >> 3;                     // workgroups per XCD (grid is a multiple of 8)
    const int q = (nt + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    const int t_end = min(nt, (xcd + 1) * q);
    int t = xcd * q + (blockIdx.x >> 3);
    if (t >= t_end) return;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;

    // DMA piece p (1 KiB = 8 rows x 128 B): lane -> row 8p + lane/8, position lane%8, logical chunk pos ^ swz(row)
    int a_row[4], a_chunk[4], b_off[NB];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a_row[i] = 8 * (4 * wave + i) + (lane >> 3);
        a_chunk[i] = 8 * ((lane & 7) ^ ((a_row[i] >> 1) & 7));
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int row = 8 * (NB * wave + i) + (lane >> 3);
        b_off[i] = row * K + 8 * ((lane & 7) ^ ((row >> 1) & 7));
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    uint32_t lds_x[4], lds_w[4];       // per-lane LDS byte addresses of the k-step fragments in buffer 0
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int fo = ((2 * ks + h) ^ ((r >> 1) & 7)) * 16;
        lds_x[ks] = lds0 + (64 * wm + r) * G_ROWB + fo;                 // token rows  (MFMA B operand)
        lds_w[ks] = lds0 + PANEL_A + (96 * wn + r) * G_ROWB + fo;       // weight rows (MFMA A operand)
    }
    const int nk = (K + G_BK - 1) / G_BK;           // K = 64 j + 32: see linear_big_kernel
    const bool ktail = (K & (G_BK - 1)) != 0;

    // per-tile DMA sources
    const bf16_t* a_src[4];
    const bf16_t* w_src;
    auto setup = [&](int tile) {
        const int64_t m0 = (int64_t)(tile / ntn) * BM;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int64_t m = m0 + a_row[i];
            m = m < M ? m : M - 1;
            a_src[i] = a + m * lda + a_chunk[i];
        }
        w_src = w + (int64_t)(tile % ntn) * G_BN * K;
    };
    auto dma = [&](int k0, int buf) {
        char* base = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t*)(a_src[i] + k0), (lptr_t*)(base + 1024 * (4 * wave + i)), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < NB; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t*)(w_src + b_off[i] + k0),
                                             (lptr_t*)(base + PANEL_A + 1024 * (NB * wave + i)), 16, 0, 0);
    };

    setup(t);
    dma(0, 0);
    int s = 0;                  // running slab counter: slab s lives in buffer s & 1
    bool trail = false;         // the previous tile left exactly NST stores behind the slab-0 DMA
    for (;;) {
        const int tn = t % ntn;
        const int64_t m0 = (int64_t)(t / ntn) * BM;
        const int n0 = tn * G_BN;
        const bool full_m = m0 + BM <= M;
        const int t_next = t + per;
        const bool has_next = t_next < t_end;

        // acc[mb][nb]: rows (registers) = n = 96*wn + 32*nb + (i&3) + 8*(i>>2) + 4*h ; column (lane) = m = 64*wm + 32*mb + r
        f32x16 acc[2][3];
        if (epilogue & MVIT_EPI_BIAS) {
#pragma unroll
            for (int nb = 0; nb < 3; ++nb) {
                const float4* bp = reinterpret_cast<const float4*>(bias + n0 + 96 * wn + 32 * nb);   // wave-uniform -> scalar loads
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {      // registers 4g..4g+3 = columns 8g + 4h + (0..3)
                    const float4 lo = bp[2 * g4], hi = bp[2 * g4 + 1];
                    const float b0 = h ? hi.x : lo.x, b1 = h ? hi.y : lo.y, b2 = h ? hi.z : lo.z, b3 = h ? hi.w : lo.w;
                    acc[0][nb][4 * g4 + 0] = b0; acc[1][nb][4 * g4 + 0] = b0;
                    acc[0][nb][4 * g4 + 1] = b1; acc[1][nb][4 * g4 + 1] = b1;
                    acc[0][nb][4 * g4 + 2] = b2; acc[1][nb][4 * g4 + 2] = b2;
                    acc[0][nb][4 * g4 + 3] = b3; acc[1][nb][4 * g4 + 3] = b3;
                }
            }
        } else {
#pragma unroll
            for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                for (int i = 0; i < 16; ++i) { acc[0][nb][i] = 0.f; acc[1][nb][i] = 0.f; }
        }

        for (int kt = 0; kt < nk; ++kt, ++s) {
            if (kt == 0 && trail) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of slab s have landed
            __builtin_amdgcn_s_barrier();                          // everyone's have; the other buffer is free
            if (kt + 1 < nk) dma((ktail && kt + 2 == nk) ? K - G_BK : (kt + 1) * G_BK, (s + 1) & 1);
            else if (has_next) { setup(t_next); dma(0, (s + 1) & 1); }
            const bool half = ktail && kt + 1 == nk;
            const uint32_t bo = (s & 1) ? BUF : 0;
            bf16x8 xf[4][2], wf[4][3];
#define RD(KS) { const uint32_t xa = lds_x[KS] + bo, wa = lds_w[KS] + bo; \
                 xf[KS][0] = lds_read128<0>(xa); xf[KS][1] = lds_read128<32 * G_ROWB>(xa); \
                 wf[KS][0] = lds_read128<0>(wa); wf[KS][1] = lds_read128<32 * G_ROWB>(wa); wf[KS][2] = lds_read128<64 * G_ROWB>(wa); }
#define MM(KS) _Pragma("unroll") for (int nb = 0; nb < 3; ++nb) _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) \
                 acc[mb][nb] = mfma16(wf[KS][nb], xf[KS][mb], acc[mb][nb]);
            // fragment pipeline: two k-steps of reads in flight ahead of the MFMAs (LDS returns in order)
            RD(0) RD(1)
            lds_wait5<5>(xf[0][0], xf[0][1], wf[0][0], wf[0][1], wf[0][2]);
            if (half) { xf[0][0] = bf16x8{}; xf[0][1] = bf16x8{}; }
            RD(2)
            MM(0)
            __builtin_amdgcn_sched_barrier(0);
            lds_wait5<5>(xf[1][0], xf[1][1], wf[1][0], wf[1][1], wf[1][2]);
            if (half) { xf[1][0] = bf16x8{}; xf[1][1] = bf16x8{}; }
            RD(3)
            MM(1)
            __builtin_amdgcn_sched_barrier(0);
            lds_wait5<5>(xf[2][0], xf[2][1], wf[2][0], wf[2][1], wf[2][2]);
            MM(2)
            __builtin_amdgcn_sched_barrier(0);
            lds_wait5<0>(xf[3][0], xf[3][1], wf[3][0], wf[3][1], wf[3][2]);
            MM(3)
#undef RD
#undef MM
        }

        // ---- epilogue from registers: lane = output row, quads of 4 consecutive columns ------------------
        auto emit = [&](auto full_tag) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int64_t m = m0 + 64 * wm + 32 * mb + r;
            const bool ok = decltype(full_tag)::value || m < M;
#pragma unroll
            for (int nb = 0; nb < 3; ++nb) {
                const int nbase = n0 + 96 * wn + 32 * nb + 4 * h;    // + 8*q
                float4 v[4];
#pragma unroll
                for (int qd = 0; qd < 4; ++qd)
                    v[qd] = make_float4(acc[mb][nb][4 * qd], acc[mb][nb][4 * qd + 1], acc[mb][nb][4 * qd + 2], acc[mb][nb][4 * qd + 3]);
                if constexpr (GELU == 2 && sizeof(TO) == 2) {       // the pre-activation first, same 16-B pieces
#pragma unroll
                    for (int qd = 0; qd < 4; qd += 2) {
                        uint32_t a0 = pack_bf16x2(v[qd].x, v[qd].y), a1 = pack_bf16x2(v[qd].z, v[qd].w);
                        uint32_t b0 = pack_bf16x2(v[qd + 1].x, v[qd + 1].y), b1 = pack_bf16x2(v[qd + 1].z, v[qd + 1].w);
                        const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                        const uint4 o = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                        if (ok) *reinterpret_cast<uint4*>(y2 + m * ldy + (n0 + 96 * wn + 32 * nb) + 8 * (qd + h)) = o;
                    }
                }
                if constexpr (GELU == 3 && sizeof(TO) == 2) {       // value and derivative from one evaluation; the derivative goes out first
                    float4 dv[4];
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        gelu_and_grad_fast(v[qd].x, v[qd].x, dv[qd].x); gelu_and_grad_fast(v[qd].y, v[qd].y, dv[qd].y);
                        gelu_and_grad_fast(v[qd].z, v[qd].z, dv[qd].z); gelu_and_grad_fast(v[qd].w, v[qd].w, dv[qd].w);
                    }
#pragma unroll
                    for (int qd = 0; qd < 4; qd += 2) {
                        uint32_t a0 = pack_bf16x2(dv[qd].x, dv[qd].y), a1 = pack_bf16x2(dv[qd].z, dv[qd].w);
                        uint32_t b0 = pack_bf16x2(dv[qd + 1].x, dv[qd + 1].y), b1 = pack_bf16x2(dv[qd + 1].z, dv[qd + 1].w);
                        const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                        const uint4 o = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                        if (ok) *reinterpret_cast<uint4*>(y2 + m * ldy + (n0 + 96 * wn + 32 * nb) + 8 * (qd + h)) = o;
                    }
                } else if (GELU) {
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        v[qd].x = gelu_fast(v[qd].x); v[qd].y = gelu_fast(v[qd].y);
                        v[qd].z = gelu_fast(v[qd].z); v[qd].w = gelu_fast(v[qd].w);
                    }
                }
                if constexpr (sizeof(TO) == 4) {
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd)
                        if (ok) *reinterpret_cast<float4*>(y + m * ldy + nbase + 8 * qd) = v[qd];
                } else {
                    // 16-bit out: pair the quads of the two half-waves with permlane32_swap -> 16-B pieces
#pragma unroll
                    for (int qd = 0; qd < 4; qd += 2) {
                        uint32_t a0 = pack_bf16x2(v[qd].x, v[qd].y), a1 = pack_bf16x2(v[qd].z, v[qd].w);
                        uint32_t b0 = pack_bf16x2(v[qd + 1].x, v[qd + 1].y), b1 = pack_bf16x2(v[qd + 1].z, v[qd + 1].w);
                        const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                        const uint4 o = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                        if (ok) *reinterpret_cast<uint4*>(y + m * ldy + (n0 + 96 * wn + 32 * nb) + 8 * (qd + h)) = o;
                    }
                }
            }
        }
        };
        if (full_m) emit(std::true_type{}); else emit(std::false_type{});
        if (!has_next) break;
        trail = full_m;     // exactly NST stores were issued behind the slab-0 DMA of the next tile
        t = t_next;
    }
}

template <typename TO, int GELU, int WM>
static int launch_linear_pers_t(const void* a, int64_t lda, const void* w, const float* bias, void* y2,
                                int64_t ldr, const float* row_scale, int64_t rps, void* y, int64_t ldy, int64_t M, int N,
                                int K, int epi, hipStream_t st) {
    constexpr int BM = 64 * WM, SMEM = 2 * (BM * G_ROWB + G_PANEL_B);
    const int64_t nt = ((M + BM - 1) / BM) * (N / G_BN);
    if (nt > 0x7fffffff) return MVIT_EINVAL;
    static DevFlags attr_done_tab; DevFlag attr_done = dev_flag(attr_done_tab);
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_pers_kernel<TO, GELU, WM>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess)
            return MVIT_ELAUNCH;
        attr_done = true;
    }
    const int64_t q = (nt + 7) / 8;
    constexpr int slots = WM == 2 ? 64 : 32;            // resident workgroups per XCD (2 or 1 per CU x 32 CUs)
    const int per = (int)(q < slots ? q : slots);
    hipLaunchKernelGGL((linear_pers_kernel<TO, GELU, WM>), dim3((unsigned)(8 * per)), dim3(128 * WM), SMEM, st, (const bf16_t*)a, lda,
                       (const bf16_t*)w, bias, (TO*)y2, ldr, row_scale, rps, (TO*)y, ldy, M, N, K, epi);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}
template <typename TO, int GELU>
static int launch_linear_pers(const void* a, int64_t lda, const void* w, const float* bias, void* y2,
                              int64_t ldr, const float* row_scale, int64_t rps, void* y, int64_t ldy, int64_t M, int N,
                              int K, int epi, hipStream_t st) {
    // Measured on the model's shapes (profiles/r2_gemm_bm256_ab.txt): the 256-row, 8-wave tile is within +-2 % of the 128-row one
    // for K = 384, 6-8 % faster for K >= 1536 with a plain epilogue (fc1 data gradient) and 6-13 % slower with the GELU
    // epilogues and on the 12,544-row stage (one workgroup per CU: nobody's MFMAs run under its epilogue).  MVIT_GEMM_BM256=1/0 forces.
    static const char* env = getenv("MVIT_GEMM_BM256");
    const bool bm256 = env ? env[0] == '1' : (GELU == 0 && K >= 1024 && M >= 32768);
    if (bm256 && 512 * lda < (1ll << 31))
        return launch_linear_pers_t<TO, GELU, 4>(a, lda, w, bias, y2, ldr, row_scale, rps, y, ldy, M, N, K, epi, st);
    return launch_linear_pers_t<TO, GELU, 2>(a, lda, w, bias, y2, ldr, row_scale, rps, y, ldy, M, N, K, epi, st);
}

template <typename TA, typename TO>
static int launch_linear_mfma(const void* a, int64_t lda, const void* w, const float* bias, const float* residual,
                              int64_t ldr, const float* row_scale, int64_t rps, void* y, int64_t ldy, int64_t M, int N,
                              int K, int epi, hipStream_t st) {
    // 256-row tiles (64x96 per wave) when that still leaves >= 2 workgroups per CU; else 128-row tiles
    const int64_t nwg2 = ((M + 2 * LBM - 1) / (2 * LBM)) * (N / LBN);
    // measured on MI355X: the 256-row tile (2 waves/SIMD) loses to the 128-row tile (3 waves/SIMD) on every shape
    // of the model with this barrier-synchronous structure; kept for the pipelined version.
    if (false && nwg2 >= 512 && sizeof(TA) == 2) {
        if (nwg2 > 0x7fffffff) return MVIT_EINVAL;
        constexpr int smem2 = 2 * LBM * L_ROWB + LBN * L_ROWB;   // 67584 >= stage (51200)
        static DevFlags attr_done_tab; DevFlag attr_done = dev_flag(attr_done_tab);
        if (!attr_done) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_mfma_kernel<TA, TO, 2>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, smem2) != hipSuccess)
                return MVIT_ELAUNCH;
            attr_done = true;
        }
        hipLaunchKernelGGL((linear_mfma_kernel<TA, TO, 2>), dim3((unsigned)nwg2), dim3(256), smem2, st, (const TA*)a, lda,
                           (const bf16_t*)w, bias, residual, ldr, row_scale, rps, (TO*)y, ldy, M, N, K, epi);
    } else {
        const int64_t nwg = ((M + LBM - 1) / LBM) * (N / LBN);
        hipLaunchKernelGGL((linear_mfma_kernel<TA, TO, 1>), dim3((unsigned)nwg), dim3(256), L_SMEM_BYTES, st, (const TA*)a,
                           lda, (const bf16_t*)w, bias, residual, ldr, row_scale, rps, (TO*)y, ldy, M, N, K, epi);
    }
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
    const bool big_shape = a_dtype == MVIT_BF16 && N % G_BN == 0 && G_KOK(K);      // the 128x192 kernels take any such K
    if ((K % LBK && !big_shape) || N % LBN || (lda & 7) || (ldy & 3) || ((epilogue & MVIT_EPI_RESIDUAL) && (ldr & 3)))
        return MVIT_EUNSUPPORTED;
#define DISPATCH(TA, TO) \
    return launch_linear_mfma<TA, TO>(a, lda, w, bias, residual, ldr, row_scale, rows_per_scale, y, ldy, M, N, K, epilogue, st)
    if (a_dtype == MVIT_BF16 && (ldy & 7) == 0) {
        const bool res = (epilogue & MVIT_EPI_RESIDUAL) != 0, gelu = (epilogue & MVIT_EPI_GELU) != 0;
        int code = -1;
        if (out_dtype == MVIT_BF16 && !res && !row_scale) code = gelu ? PP_GELU16 : PP_B16;
        if (out_dtype == MVIT_F32 && !gelu) {
            if (res && (ldr & 3) == 0) code = row_scale ? PP_F32_RES_SC : PP_F32_RES;
            else if (!res && !row_scale) code = PP_F32;
        }
        if ((code == PP_B16 || code == PP_GELU16) && (epilogue & MVIT_EPI_BIAS) && mvit_internal_linear_k96_ok(lda, M, N, K))
            return mvit_internal_linear_k96(code, a, lda, w, bias, y, nullptr, ldy, M, N, st);
        if (code >= 0 && use_pp(lda, M, N, K, code))
            return mvit_internal_linear_pp(code, a, lda, w, (epilogue & MVIT_EPI_BIAS) ? bias : nullptr, residual, ldr, row_scale,
                                           rows_per_scale, y, nullptr, ldy, M, N, K, st);
    }
    static const bool use_dma = getenv("MVIT_GEMM_NO_DMA") == nullptr;
    static const bool use_big = getenv("MVIT_GEMM_NO_BIG") == nullptr;
    if (a_dtype == MVIT_BF16 && use_big && N % G_BN == 0 && G_KOK(K) && 256 * lda < (1ll << 31) && (int64_t)N * K < (1ll << 31) &&
        !(out_dtype == MVIT_BF16 && (epilogue & MVIT_EPI_RESIDUAL))) {
        // persistent form only where the epilogue has no vector loads (measured: with residual / drop-path loads the
        // one-tile-per-workgroup form overlaps them better)
        static const bool pers_env = getenv("MVIT_GEMM_NO_PERS") == nullptr;
        const bool use_pers = pers_env && !row_scale && !(epilogue & MVIT_EPI_RESIDUAL);
#define PERS(TO, G) return launch_linear_pers<TO, G>(a, lda, w, bias, nullptr, ldr, row_scale, rows_per_scale, y, ldy, M, N, K, epilogue, st)
        if (use_pers && out_dtype == MVIT_BF16) { if (epilogue & MVIT_EPI_GELU) PERS(bf16_t, 1); else PERS(bf16_t, 0); }
        if (use_pers && out_dtype == MVIT_F32) { if (epilogue & MVIT_EPI_GELU) PERS(float, 1); else PERS(float, 0); }
#undef PERS
        if (out_dtype == MVIT_BF16)
            return launch_linear_big<bf16_t>(a, lda, w, bias, residual, ldr, row_scale, rows_per_scale, y, ldy, M, N, K, epilogue, st);
        if (out_dtype == MVIT_F32)
            return launch_linear_big<float>(a, lda, w, bias, residual, ldr, row_scale, rows_per_scale, y, ldy, M, N, K, epilogue, st);
    }
    if (K % LBK) return MVIT_EUNSUPPORTED;          // the kernels below walk K in 48- / 96-wide slabs
    if (a_dtype == MVIT_BF16 && use_dma && 128 * lda < (1ll << 31)) {
        if (out_dtype == MVIT_BF16)
            return launch_linear_dma<bf16_t>(a, lda, w, bias, residual, ldr, row_scale, rows_per_scale, y, ldy, M, N, K, epilogue, st);
        if (out_dtype == MVIT_F32)
            return launch_linear_dma<float>(a, lda, w, bias, residual, ldr, row_scale, rows_per_scale, y, ldy, M, N, K, epilogue, st);
    }
    if (a_dtype == MVIT_BF16 && out_dtype == MVIT_BF16) DISPATCH(bf16_t, bf16_t);
    if (a_dtype == MVIT_BF16 && out_dtype == MVIT_F32) DISPATCH(bf16_t, float);
    if (a_dtype == MVIT_F32 && out_dtype == MVIT_F32) DISPATCH(float, float);
    if (a_dtype == MVIT_F32 && out_dtype == MVIT_BF16) DISPATCH(float, bf16_t);
#undef DISPATCH
    return MVIT_EDTYPE;
}

// fc1 of the MLP in a training step: pre = a . w^T + bias and y = GELU(pre) from ONE pass over the accumulators (the backward
// needs pre, the forward continues with y).  16-bit operands / outputs; shapes the persistent kernel does not cover run the
// plain GEMM followed by the element-wise kernel.
extern "C" int mvit_gelu_fwd(const void* x, void* y, int64_t n, int act_dtype, void* stream);
extern "C" int mvit_linear_gelu_fwd(const void* a, int64_t lda, const void* w, const float* bias, void* pre, void* y, int64_t M,
                                    int N, int K, int act_dtype, void* stream) {
    if (!a || !w || !bias || !pre || !y || M < 0 || N <= 0 || K <= 0) return MVIT_EINVAL;
    if (act_dtype != MVIT_BF16) return MVIT_EDTYPE;
    if (M == 0) return MVIT_OK;
    static const bool fused = getenv("MVIT_GEMM_NO_PERS") == nullptr && getenv("MVIT_GEMM_NO_BIG") == nullptr && getenv("MVIT_NO_GELU_FUSE") == nullptr;
    if (fused && (lda & 7) == 0 && mvit_internal_linear_k96_ok(lda, M, N, K))
        return mvit_internal_linear_k96(PP_GELU_PRE, a, lda, w, bias, y, pre, N, M, N, as_stream(stream));
    if (fused && use_pp(lda, M, N, K, PP_GELU_PRE))
        return mvit_internal_linear_pp(PP_GELU_PRE, a, lda, w, bias, nullptr, 0, nullptr, 0, y, pre, N, M, N, K, as_stream(stream));
    if (fused && N % G_BN == 0 && G_KOK(K) && (lda & 7) == 0 && 256 * lda < (1ll << 31) && (int64_t)N * K < (1ll << 31))
        return launch_linear_pers<bf16_t, 2>(a, lda, w, bias, pre, 0, nullptr, 0, y, N, M, N, K, MVIT_EPI_BIAS, as_stream(stream));
    const int rc = mvit_linear_fwd(a, MVIT_BF16, lda, w, bias, nullptr, 0, nullptr, 0, pre, MVIT_BF16, N, M, N, K, MVIT_EPI_BIAS, act_dtype, stream);
    if (rc != MVIT_OK) return rc;
    return mvit_gelu_fwd(pre, y, M * (int64_t)N, act_dtype, stream);
}

// Data gradient of fc2 fused with the GELU backward: y = GELU'(pre) * row_scale[m/rps] * (a . w^T), i.e. the gradient of fc1's
// output straight from the GEMM (a = d_out rows in 16 bit, w = fc2.weight^T [N][K], pre / y [M][N] 16-bit).  Shapes the 128x192
// kernel does not cover run the plain GEMM followed by the element-wise kernel.
extern "C" int mvit_gelu_bwd(const void* x, const void* dy, void* dx, int64_t n, int act_dtype, void* stream);
extern "C" int mvit_linear_dgelu_fwd(const void* a, int64_t lda, const void* w, const float* row_scale, int64_t rows_per_scale,
                                     const void* pre, void* y, int64_t M, int N, int K, int act_dtype, void* stream) {
    if (!a || !w || !pre || !y || M < 0 || N <= 0 || K <= 0 || (row_scale && rows_per_scale <= 0)) return MVIT_EINVAL;
    if (act_dtype != MVIT_BF16) return MVIT_EDTYPE;
    if (M == 0) return MVIT_OK;
    static const bool fused = getenv("MVIT_GEMM_NO_BIG") == nullptr && getenv("MVIT_NO_GELU_FUSE") == nullptr;
    if (fused && use_pp(lda, M, N, K, PP_DG_PRE))
        return mvit_internal_linear_pp(PP_DG_PRE, a, lda, w, nullptr, pre, N, row_scale, rows_per_scale, y, nullptr, N, M, N, K, as_stream(stream));
    if (fused && N % G_BN == 0 && G_KOK(K) && (lda & 7) == 0 && 256 * lda < (1ll << 31) && (int64_t)N * K < (1ll << 31))
        return launch_linear_big_t<bf16_t, false, false, 1>(a, lda, w, nullptr, reinterpret_cast<const float*>(pre), N, row_scale,
                                                            rows_per_scale, y, N, M, N, K, 0, as_stream(stream));
    const int rc = mvit_linear_fwd(a, MVIT_BF16, lda, w, nullptr, nullptr, 0, row_scale, rows_per_scale, y, MVIT_BF16, N, M, N, K, 0,
                                   act_dtype, stream);
    if (rc != MVIT_OK) return rc;
    return mvit_gelu_bwd(pre, y, y, M * (int64_t)N, act_dtype, stream);
}

// The same pair with the derivative kept instead of the pre-activation: mvit_linear_gelu_fwd_dsave writes y = GELU(pre) and
// dact = GELU'(pre) (one erf / exp evaluation for both); mvit_linear_dact_fwd multiplies the fc2 data gradient by that saved
// factor (no transcendental in the backward epilogue).  Only for shapes of the 128x192 kernels (N % 192 == 0, K % 64 == 0): they
// return MVIT_EUNSUPPORTED otherwise and the caller uses the pre-activation pair above.
extern "C" int mvit_linear_gelu_fwd_dsave(const void* a, int64_t lda, const void* w, const float* bias, void* dact, void* y, int64_t M,
                                          int N, int K, int act_dtype, void* stream) {
    if (!a || !w || !bias || !dact || !y || M < 0 || N <= 0 || K <= 0) return MVIT_EINVAL;
    if (act_dtype != MVIT_BF16) return MVIT_EDTYPE;
    if (M == 0) return MVIT_OK;
    if (mvit_internal_linear_k96_ok(lda, M, N, K))
        return mvit_internal_linear_k96(PP_GELU_DER, a, lda, w, bias, y, dact, N, M, N, as_stream(stream));
    if (N % G_BN || !G_KOK(K) || (lda & 7) || 256 * lda >= (1ll << 31) || (int64_t)N * K >= (1ll << 31)) return MVIT_EUNSUPPORTED;
    if (use_pp(lda, M, N, K, PP_GELU_DER))
        return mvit_internal_linear_pp(PP_GELU_DER, a, lda, w, bias, nullptr, 0, nullptr, 0, y, dact, N, M, N, K, as_stream(stream));
    return launch_linear_pers<bf16_t, 3>(a, lda, w, bias, dact, 0, nullptr, 0, y, N, M, N, K, MVIT_EPI_BIAS, as_stream(stream));
}
extern "C" int mvit_linear_dact_fwd(const void* a, int64_t lda, const void* w, const float* row_scale, int64_t rows_per_scale,
                                    const void* dact, void* y, int64_t M, int N, int K, int act_dtype, void* stream) {
    if (!a || !w || !dact || !y || M < 0 || N <= 0 || K <= 0 || (row_scale && rows_per_scale <= 0)) return MVIT_EINVAL;
    if (act_dtype != MVIT_BF16) return MVIT_EDTYPE;
    if (M == 0) return MVIT_OK;
    if (N % G_BN || !G_KOK(K) || (lda & 7) || 256 * lda >= (1ll << 31) || (int64_t)N * K >= (1ll << 31)) return MVIT_EUNSUPPORTED;
    if (use_pp(lda, M, N, K, PP_DG_DER))
        return mvit_internal_linear_pp(PP_DG_DER, a, lda, w, nullptr, dact, N, row_scale, rows_per_scale, y, nullptr, N, M, N, K, as_stream(stream));
    return launch_linear_big_t<bf16_t, false, false, 2>(a, lda, w, nullptr, reinterpret_cast<const float*>(dact), N, row_scale,
                                                        rows_per_scale, y, N, M, N, K, 0, as_stream(stream));
}
