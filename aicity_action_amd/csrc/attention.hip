// Fused pooled attention, head_dim 96:  O = softmax(q k^T * scale) v (+ q), scores never leave the CU.
//
// bf16 path (MFMA 32x32x16): one wave = 32 queries, 4 waves per workgroup, 64-key tiles.
//   S^T = K . Q^T   (A = K rows from LDS via ds_read_b128, B = Q^T held in registers) puts a QUERY on
//   each lane and 16 keys in its accumulator registers, so the softmax row reductions are in-register
//   (+ one cross-half exchange) and the exponentiated tile is already the B operand of
//   O^T += V^T . P^T (A = V^T read from a row-major V tile with ds_read_b64_tr_b16).
// fp32 path: exact-fp32 VALU kernel, one query per thread (parity path).
#include "common.h"

#define A_KT 64           // keys per tile
#define A_ROWB 192        // bytes per K/V row in LDS (96 bf16)
#define A_QW 32           // queries per wave
#define A_QB 128          // queries per workgroup

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ int kslab_off(int row, int chunk) {
    int p = chunk + ((row >> 2) & 3);
    p = p >= 12 ? p - 12 : p;
    return row * A_ROWB + p * 16;
}

template <bool ADD_Q>
__global__ __launch_bounds__(256) void attn_fwd_bf16_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kt,
                                                            const bf16_t* __restrict__ V, bf16_t* __restrict__ O,
                                                            float* __restrict__ LSE, int heads, int Lq, int Lk,
                                                            float scale_log2e) {
    __shared__ __attribute__((aligned(16))) char smem[2 * A_KT * A_ROWB];
    char* sK = smem;
    char* sV = smem + A_KT * A_ROWB;

    const int bh = blockIdx.y;            // b*heads + g
    const int b = bh / heads, g = bh - b * heads;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int q0 = blockIdx.x * A_QB + wave * A_QW;

    const bf16_t* Qb = Q + (int64_t)bh * Lq * 96;
    const bf16_t* Kb = Kt + (int64_t)bh * Lk * 96;
    const bf16_t* Vb = V + (int64_t)bh * Lk * 96;

    // Q^T fragments: lane (r,h) holds Q[q0+r][16ks + 8h .. +7]
    int qi = q0 + r;
    const bool q_ok = qi < Lq;
    qi = q_ok ? qi : Lq - 1;
    bf16x8 qf[6];
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(Qb + (int64_t)qi * 96 + 16 * ks + 8 * h);

    // staging map: a 64-key tile is one contiguous 12 KiB block of K (and of V); thread tid moves the 16-byte chunks
    // c = tid + 256*i.  Global side: uniform tile base (scalar) + a constant 32-bit per-thread offset, so the loop
    // spends no vector instructions on addresses.
    int s_koff[3], s_voff[3];
    uint32_t g_off[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = tid + 256 * i;
        const int row = c / 12, chk = c - row * 12;
        s_koff[i] = kslab_off(row, chk);
        s_voff[i] = c * 16;
        g_off[i] = (uint32_t)c * 16u;
    }
    uint4 rk[3], rv[3];
    auto gload = [&](int k0) {
        const char* kt_base = reinterpret_cast<const char*>(Kb) + (int64_t)k0 * A_ROWB;   // wave-uniform
        const char* vt_base = reinterpret_cast<const char*>(Vb) + (int64_t)k0 * A_ROWB;
        if (k0 + A_KT <= Lk) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                rk[i] = *reinterpret_cast<const uint4*>(kt_base + g_off[i]);
                rv[i] = *reinterpret_cast<const uint4*>(vt_base + g_off[i]);
            }
        } else {
            const uint32_t lim = (uint32_t)(Lk - k0) * A_ROWB;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (g_off[i] < lim) {
                    rk[i] = *reinterpret_cast<const uint4*>(kt_base + g_off[i]);
                    rv[i] = *reinterpret_cast<const uint4*>(vt_base + g_off[i]);
                } else {
                    rk[i] = make_uint4(0, 0, 0, 0);
                    rv[i] = make_uint4(0, 0, 0, 0);
                }
            }
        }
    };

    int koff[6];  // K fragment chunk offsets for this lane
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
        int p = 2 * ks + h + ((r >> 2) & 3);
        p = p >= 12 ? p - 12 : p;
        koff[ks] = p * 16;
    }
    // transposed V read: lane supplies the address of key-row (4h + q) and d columns 16*(gi&1) + 4p
    const int i16 = lane & 15, gi = lane >> 4;
    const int v_lane_off = (4 * h + (i16 >> 2)) * A_ROWB + (16 * (gi & 1) + 4 * (i16 & 3)) * 2;

    f32x16 o[3];
#pragma unroll
    for (int db = 0; db < 3; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int nkt = (Lk + A_KT - 1) / A_KT;
    gload(0);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            *reinterpret_cast<uint4*>(sK + s_koff[i]) = rk[i];
            *reinterpret_cast<uint4*>(sV + s_voff[i]) = rv[i];
        }
        __syncthreads();
        if (kt + 1 < nkt) gload((kt + 1) * A_KT);

        // ---- S^T = K . Q^T  (two 32-key blocks) ------------------------------------------------
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {       // the two key blocks alternate: no back-to-back dependent MFMAs
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + (32 * kb + r) * A_ROWB + koff[ks]);
                s[kb] = mfma16(kf, qf[ks], s[kb]);
            }
        }
        // ---- online softmax: raw-score running max, scale folded into the exp2 argument -----------------
        const int kbase = kt * A_KT;
        if (kbase + A_KT > Lk) {           // tail tile only (wave-uniform): mask keys >= Lk
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kbase + 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * h;
                    s[kb][i] = key < Lk ? s[kb][i] : -INFINITY;
                }
        }
        float mx = fmaxf(s[0][0], s[1][0]);
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(fmaxf(mx, s[0][i]), s[1][i]);      // v_max3_f32
        {   // both halves of the wave hold the same 32 queries (different keys): exchange maxima
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        const float m_new = fmaxf(m_run, mx);
        if (__any(m_new > m_run)) {        // rescale only when some query's running max moved
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);
            l_run *= alpha;
#pragma unroll
            for (int db = 0; db < 3; ++db)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[db][i] *= alpha;
            m_run = m_new;
        }
        const f32x2 c2 = {scale_log2e, scale_log2e};
        const float mcs = -m_run * scale_log2e;
        const f32x2 mc2 = {mcs, mcs};
        f32x2 ps2 = {0.f, 0.f};
        bf16x8 pf[4];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
                uint32_t pk[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const f32x2 sv = {s[kb][8 * sh + 2 * jj], s[kb][8 * sh + 2 * jj + 1]};
                    const f32x2 t = __builtin_elementwise_fma(sv, c2, mc2);            // v_pk_fma_f32
                    const f32x2 pp = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                    ps2 += pp;                                                          // v_pk_add_f32
                    pk[jj] = pack_bf16x2(pp[0], pp[1]);
                }
                uint4 u = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                pf[2 * kb + sh] = *reinterpret_cast<bf16x8*>(&u);
            }
        const float psum = ps2[0] + ps2[1];
        l_run += psum;
        // ---- O^T += V^T . P^T -------------------------------------------------------------------
#pragma unroll
        for (int s16 = 0; s16 < 4; ++s16) {
#pragma unroll
            for (int db = 0; db < 3; ++db) {
                const char* vp = sV + v_lane_off + s16 * 16 * A_ROWB + db * 64;
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(vp));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(vp + 8 * A_ROWB));
                bf16x8 vf;
                vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                o[db] = mfma16(vf, pf[s16], o[db]);
            }
        }
    }
    // ---- epilogue: normalise, + q residual, store [b][q][g*96 + d] -------------------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (LSE && q_ok && h == 0) LSE[(int64_t)bh * Lq + qi] = m_run * scale_log2e + __builtin_amdgcn_logf(l_tot);  // log2 domain
    if (q_ok) {
        const int C = heads * 96;
        bf16_t* orow = O + ((int64_t)b * Lq + qi) * C + g * 96;
        const bf16_t* qrow = Qb + (int64_t)qi * 96;
#pragma unroll
        for (int db = 0; db < 3; ++db)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const int d = 32 * db + 8 * i4 + 4 * h;
                float4 v = make_float4(o[db][4 * i4 + 0] * inv, o[db][4 * i4 + 1] * inv, o[db][4 * i4 + 2] * inv,
                                       o[db][4 * i4 + 3] * inv);
                if (ADD_Q) {
                    const float4 qq = load4(qrow + d);
                    v.x += qq.x; v.y += qq.y; v.z += qq.z; v.w += qq.w;
                }
                store4(orow + d, v);
            }
    }
}

// ------------------------------------------------------------------------------------------------
// exact fp32 path: one query per thread, 32-key tiles in LDS (broadcast reads).
// ------------------------------------------------------------------------------------------------
#define F_KT 32
__global__ __launch_bounds__(128) void attn_fwd_f32_kernel(const float* __restrict__ Q, const float* __restrict__ Kt,
                                                           const float* __restrict__ V, float* __restrict__ O,
                                                           float* __restrict__ LSE, int heads, int Lq, int Lk, float scale,
                                                           int add_q) {
    __shared__ __attribute__((aligned(16))) float sK[F_KT * 96];
    __shared__ __attribute__((aligned(16))) float sV[F_KT * 96];
    const int bh = blockIdx.y;
    const int b = bh / heads, g = bh - b * heads;
    int qi = blockIdx.x * 128 + threadIdx.x;
    const bool q_ok = qi < Lq;
    qi = q_ok ? qi : Lq - 1;
    const float* qrow = Q + ((int64_t)bh * Lq + qi) * 96;
    float q[96], o[96];
#pragma unroll
    for (int d = 0; d < 96; d += 4) {
        const float4 v = load4(qrow + d);
        q[d] = v.x; q[d + 1] = v.y; q[d + 2] = v.z; q[d + 3] = v.w;
        o[d] = o[d + 1] = o[d + 2] = o[d + 3] = 0.f;
    }
    float m_run = -INFINITY, l_run = 0.f;
    const float* Kb = Kt + (int64_t)bh * Lk * 96;
    const float* Vb = V + (int64_t)bh * Lk * 96;
    for (int k0 = 0; k0 < Lk; k0 += F_KT) {
        __syncthreads();
        for (int i = threadIdx.x; i < F_KT * 24; i += 128) {
            const int row = i / 24, c4 = i - row * 24;
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
            if (k0 + row < Lk) {
                kv = load4(Kb + (int64_t)(k0 + row) * 96 + 4 * c4);
                vv = load4(Vb + (int64_t)(k0 + row) * 96 + 4 * c4);
            }
            *reinterpret_cast<float4*>(sK + row * 96 + 4 * c4) = kv;
            *reinterpret_cast<float4*>(sV + row * 96 + 4 * c4) = vv;
        }
        __syncthreads();
        const int nk = min(F_KT, Lk - k0);
        float sc[F_KT];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < F_KT; ++j) {
            float a = 0.f;
#pragma unroll
            for (int d = 0; d < 96; d += 4) {
                const float4 kv = *reinterpret_cast<const float4*>(sK + j * 96 + d);
                a = fmaf(q[d], kv.x, a); a = fmaf(q[d + 1], kv.y, a);
                a = fmaf(q[d + 2], kv.z, a); a = fmaf(q[d + 3], kv.w, a);
            }
            a = (j < nk) ? a * scale : -INFINITY;
            sc[j] = a;
            mx = fmaxf(mx, a);
        }
        const float m_new = fmaxf(m_run, mx);
        const float alpha = expf(m_run - m_new);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int d = 0; d < 96; ++d) o[d] *= alpha;
#pragma unroll
        for (int j = 0; j < F_KT; ++j) {
            const float p = expf(sc[j] - m_new);
            l_run += p;
#pragma unroll
            for (int d = 0; d < 96; d += 4) {
                const float4 vv = *reinterpret_cast<const float4*>(sV + j * 96 + d);
                o[d] = fmaf(p, vv.x, o[d]); o[d + 1] = fmaf(p, vv.y, o[d + 1]);
                o[d + 2] = fmaf(p, vv.z, o[d + 2]); o[d + 3] = fmaf(p, vv.w, o[d + 3]);
            }
        }
    }
    if (q_ok) {
        const float inv = 1.0f / l_run;
        if (LSE) LSE[(int64_t)bh * Lq + qi] = (m_run + logf(l_run)) * 1.44269504088896340736f;   // log2 units
        float* orow = O + ((int64_t)b * Lq + qi) * (heads * 96) + g * 96;
#pragma unroll
        for (int d = 0; d < 96; d += 4) {
            float4 v = make_float4(o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv);
            if (add_q) { v.x += q[d]; v.y += q[d + 1]; v.z += q[d + 2]; v.w += q[d + 3]; }
            store4(orow + d, v);
        }
    }
}

extern "C" int mvit_attention_fwd(const void* q, const void* k, const void* v, void* out, float* lse, int B, int heads,
                                  int Lq, int Lk, float scale, int add_q, int act_dtype, void* stream) {
    if (!q || !k || !v || !out || B <= 0 || heads <= 0 || Lq <= 0 || Lk <= 0) return MVIT_EINVAL;
    hipStream_t st = as_stream(stream);
    if ((int64_t)B * heads > 65535) return MVIT_EINVAL;
    if (act_dtype == MVIT_BF16) {
        dim3 grid((Lq + A_QB - 1) / A_QB, B * heads);
        const float sl2 = scale * 1.44269504088896340736f;
        if (add_q)
            hipLaunchKernelGGL((attn_fwd_bf16_kernel<true>), grid, dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)k,
                               (const bf16_t*)v, (bf16_t*)out, lse, heads, Lq, Lk, sl2);
        else
            hipLaunchKernelGGL((attn_fwd_bf16_kernel<false>), grid, dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)k,
                               (const bf16_t*)v, (bf16_t*)out, lse, heads, Lq, Lk, sl2);
    } else if (act_dtype == MVIT_F32) {
        dim3 grid((Lq + 127) / 128, B * heads);
        hipLaunchKernelGGL(attn_fwd_f32_kernel, grid, dim3(128), 0, st, (const float*)q, (const float*)k, (const float*)v,
                           (float*)out, lse, heads, Lq, Lk, scale, add_q);
    } else {
        return MVIT_EDTYPE;
    }
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}
