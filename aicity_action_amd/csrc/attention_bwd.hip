// Backward of the fused pooled attention (head_dim 96), recompute-from-LSE, no score matrix, NO float atomics:
//   delta[q]  = sum_d dO[q][d] * O_attn[q][d]                               (attn_bwd_delta_kernel)
//   pass A    : one wave = 32 queries, loops over 64-key tiles:  dQ  = scale * sum_k dS[q][k] K[k]   (+ dO if the
//               forward added the pooled-q residual)
//   pass B    : one wave = 32 keys, loops over 64-query tiles:   dV  = sum_q P[q][k] dO[q],  dK = scale * sum_q dS[q][k] Q[q]
//   with P = exp2(score*scale*log2e - LSE2[q]),  dS = P * (dP - delta[q]),  dP = dO . V^T.
// Both passes are the forward kernel's skeleton (resident operand as MFMA B fragments in registers, streamed tiles in
// LDS, accumulator tile reused directly as the B operand of the second product); recomputing S/dP in both passes
// costs 7 instead of 5 matrix products but keeps the pass deterministic and free of the 1.3 TB/s atomic ceiling.
#include <type_traits>

#include "common.h"

#ifndef BWD_ABL
#define BWD_ABL 0         // timing ablations of the dQ pass (results invalid): 1 no dO re-read in the epilogue, 2 no dQ stores, 4 no register operand loads, 8 dK/dV pass without its lse / delta reads, 16 without its exponential arithmetic, 32 without fragment reads, 64 without MFMAs
#endif
#define B_T 64            // streamed rows per tile
#define B_ROWB 192
typedef __attribute__((address_space(3))) bf16x4 lds_b4;

__device__ __forceinline__ int rot_off(int row, int chunk) {   // rotation-swizzled [rows][96] bf16 image (row reads)
    int p = chunk + ((row >> 2) & 3);
    p = p >= 12 ? p - 12 : p;
    return row * B_ROWB + p * 16;
}

__device__ __forceinline__ bf16x8 tr_frag(const char* p) {      // two transposed 4x16 reads -> one 8-element fragment
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b4*)(p));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b4*)(p + 8 * B_ROWB));
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

__device__ __forceinline__ bf16x8 pack8(const float* v) {
    uint4 u = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
    return *reinterpret_cast<bf16x8*>(&u);
}

// ------------------------------------------------------------------------------------------------
// delta: 8 lanes per (b, g, q), 12 channels each.  dO, O: [B][Lq][heads*96]; Q: [B][heads][Lq][96].
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_delta_kernel(const T* __restrict__ dO, const T* __restrict__ O,
                                                             const T* __restrict__ Q, float* __restrict__ delta, int B,
                                                             int heads, int Lq, int add_q) {
    // 4 lanes per (b, g, q) row; lane j owns the 16-byte chunks j, j+4, ... of the row's 96 channels
    constexpr int CW = 16 / sizeof(T);          // channels per 16-byte chunk
    constexpr int NCH = 96 / CW / 4;            // chunks per lane
    const int64_t total = (int64_t)B * heads * Lq;
    const int j = threadIdx.x & 3;
    for (int64_t it0 = (int64_t)blockIdx.x * 64; it0 < total; it0 += (int64_t)gridDim.x * 64) {
        const int64_t it = it0 + (threadIdx.x >> 2);
        const bool ok = it < total;
        const int64_t itc = ok ? it : total - 1;
        const int q = (int)(itc % Lq);
        const int g = (int)((itc / Lq) % heads);
        const int b = (int)(itc / ((int64_t)Lq * heads));
        const int64_t orow = ((int64_t)b * Lq + q) * heads * 96 + g * 96;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c0 = CW * (j + 4 * i);
            float4 d0, d1, o0, o1;
            if constexpr (CW == 8) {
                load8(dO + orow + c0, d0, d1);
                load8(O + orow + c0, o0, o1);
                if (add_q) {
                    float4 q0, q1;
                    load8(Q + itc * 96 + c0, q0, q1);
                    o0.x -= q0.x; o0.y -= q0.y; o0.z -= q0.z; o0.w -= q0.w;
                    o1.x -= q1.x; o1.y -= q1.y; o1.z -= q1.z; o1.w -= q1.w;
                }
                s += (d0.x * o0.x + d0.y * o0.y) + (d0.z * o0.z + d0.w * o0.w) + (d1.x * o1.x + d1.y * o1.y) + (d1.z * o1.z + d1.w * o1.w);
            } else {
                d0 = load4(dO + orow + c0);
                o0 = load4(O + orow + c0);
                if (add_q) {
                    const float4 q0 = load4(Q + itc * 96 + c0);
                    o0.x -= q0.x; o0.y -= q0.y; o0.z -= q0.z; o0.w -= q0.w;
                }
                s += (d0.x * o0.x + d0.y * o0.y) + (d0.z * o0.z + d0.w * o0.w);
            }
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        if (ok && j == 0) delta[it] = s;
    }
}

// 16-bit form: the [B][Lq][heads*96] tensors are walked as one flat stream of 16-byte chunks (thread = chunk, fully coalesced:
// 192 threads x 16 B = 3 KiB contiguous per load; the 4-lanes-per-row kernel above touched sixteen 64-byte pieces of sixteen rows
// per wave-instruction and ran at 2.7 TB/s); the twelve chunk sums of a (b, q, head) meet in LDS.  192 = 16 x 12 threads and
// every row is a multiple of twelve chunks, so a group never straddles a workgroup or a row.
// Qs (optional): the pre-scaled 16-bit queries round16(q * scale_log2e) -- bit for bit what the 64-query forward kernel and the dQ pass
// build in registers -- for the dK/dV pass, whose query tiles go from memory to LDS without passing through registers.
__global__ __launch_bounds__(192) void attn_bwd_delta_flat_kernel(const bf16_t* __restrict__ dO, const bf16_t* __restrict__ O,
                                                                  const bf16_t* __restrict__ Q, float* __restrict__ delta,
                                                                  int heads, int Lq, int64_t nchunks, int add_q,
                                                                  bf16_t* __restrict__ Qs, float scale_log2e) {
    __shared__ float part[192];
    const int cpr = 12 * heads;                                    // chunks per (b, q) row
    for (int64_t c0 = (int64_t)blockIdx.x * 192; c0 < nchunks; c0 += (int64_t)gridDim.x * 192) {
        const int64_t c = c0 + threadIdx.x;
        float s = 0.f;
        int64_t it = 0;
        if (c < nchunks) {
            const int64_t row = c / cpr;                           // b * Lq + q
            const int cin = (int)(c - row * cpr), g = cin / 12, k = cin - 12 * g;
            const int64_t b = row / Lq, q = row - b * Lq;
            it = (b * heads + g) * Lq + q;
            float4 d0, d1, o0, o1;
            load8(dO + c * 8, d0, d1);
            load8(O + c * 8, o0, o1);
            if (add_q || Qs) {
                float4 q0, q1;
                load8(Q + it * 96 + 8 * k, q0, q1);
                if (Qs)
                    *reinterpret_cast<uint4*>(Qs + it * 96 + 8 * k) =
                        make_uint4(pack_bf16x2(q0.x * scale_log2e, q0.y * scale_log2e), pack_bf16x2(q0.z * scale_log2e, q0.w * scale_log2e),
                                   pack_bf16x2(q1.x * scale_log2e, q1.y * scale_log2e), pack_bf16x2(q1.z * scale_log2e, q1.w * scale_log2e));
                if (add_q) {
                    o0.x -= q0.x; o0.y -= q0.y; o0.z -= q0.z; o0.w -= q0.w;
                    o1.x -= q1.x; o1.y -= q1.y; o1.z -= q1.z; o1.w -= q1.w;
                }
            }
            s = (d0.x * o0.x + d0.y * o0.y) + (d0.z * o0.z + d0.w * o0.w) + (d1.x * o1.x + d1.y * o1.y) + (d1.z * o1.z + d1.w * o1.w);
        }
        part[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x < 16) {
            const int64_t cg = c0 + 12 * threadIdx.x;              // first chunk of this group
            if (cg < nchunks) {
                const float* p = part + 12 * threadIdx.x;
                const float t = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7])) + ((p[8] + p[9]) + (p[10] + p[11]));
                const int64_t row = cg / cpr;
                const int g = (int)(cg - row * cpr) / 12;
                const int64_t b = row / Lq, q = row - b * Lq;
                delta[(b * heads + g) * Lq + q] = t;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// pass A (bf16 MFMA): dQ.   grid (ceil(Lq/128), B*heads), 4 waves x 32 queries.
// ------------------------------------------------------------------------------------------------
// K and V tiles arrive as in the forward kernel: a 3-stage LDS ring of (K image | V image) filled by
// global_load_lds_dwordx4 with the rotation swizzle on the source address (two tiles in flight, one s_barrier per tile).
// The K image serves both the row reads (S^T = K Q^T) and the transposing reads (dQ^T += K^T dS^T): the latter apply the
// row's rotation to their own address and are issued as inline asm (no compiler vmcnt(0) in front of them).

typedef __attribute__((address_space(1))) const void b_gptr_t;
typedef __attribute__((address_space(3))) void b_lptr_t;
#define BQ_STAGES 3
#define BQ_TILEB (2 * B_T * B_ROWB)      // 24 KiB

template <int OFF>
__device__ __forceinline__ bf16x4 b_tr16(uint32_t addr) {
    bf16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ bf16x8 b_rd128(uint32_t addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
__device__ __forceinline__ bf16x8 b_join(bf16x4 lo, bf16x4 hi) {
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}
#define B_WAIT6(A, N) \
    asm volatile("s_waitcnt lgkmcnt(%12)" \
                 : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]), "+v"(A[5]), "+v"(A[6]), "+v"(A[7]), "+v"(A[8]), "+v"(A[9]), \
                   "+v"(A[10]), "+v"(A[11]) \
                 : "n"(N))

// PRE: the forward of this shape ran on round16(q * scale * log2e) (attn_fwd_prescales_q): rebuild exactly those scores.  Otherwise
// the forward's scores were (q . k) * scale * log2e on the unrounded q, and so are these (one multiply-add per element more).
// FD (round 5): this pass also PRODUCES delta and the pre-scaled queries instead of reading delta: it holds its query's dO half-row in
// registers anyway, so it loads the matching half-row of the forward output, forms delta = sum_d dO (O - q) itself (two lanes per query:
// one v_add across lane ^ 32) and writes delta and round16(q * scale_log2e) for the dK/dV pass behind it on the same stream.  The
// separate delta kernel (a read of dO, O, q and a write of the scaled q: 154 MB per stage-3 call, 35 us) is then not launched.
template <bool ADD_Q, bool PRE = true, bool FD = false>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kt,
                                                             const bf16_t* __restrict__ V, const bf16_t* __restrict__ dO,
                                                             const float* __restrict__ LSE, const float* __restrict__ delta,
                                                             bf16_t* __restrict__ dQ, int heads, int Lq, int Lk, float scale,
                                                             float scale_log2e, const bf16_t* __restrict__ O = nullptr,
                                                             float* __restrict__ delta_out = nullptr, bf16_t* __restrict__ Qs = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // BQ_STAGES x (K rotation image | V rotation image)
    int qtile, bh;
    xcd_group_map(qtile, bh);
    const int b = bh / heads, g = bh - b * heads;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    int qi = qtile * 128 + wave * 32 + r;
    const bool q_ok = qi < Lq;
    qi = q_ok ? qi : Lq - 1;
    const int C = heads * 96;
    const bf16_t* Qb = Q + (int64_t)bh * Lq * 96;
    const bf16_t* Kb = Kt + (int64_t)bh * Lk * 96;
    const bf16_t* Vb = V + (int64_t)bh * Lk * 96;
    const bf16_t* dOrow = dO + ((int64_t)b * Lq + qi) * C + g * 96;

    // DMA pieces: LDS position p = 64*piece + lane holds chunk (p%12 - rot(row)) of row p/12 (K and V alike)
    uint32_t g_off[3];
    int p_row[3], p_c[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int p = 64 * (3 * wave + i) + lane;
        const int row = p / 12, pos = p - row * 12;
        int c = pos - ((row >> 2) & 3);
        c = c < 0 ? c + 12 : c;
        p_row[i] = row; p_c[i] = c;
        g_off[i] = (uint32_t)(row * 12 + c) * 16u;
    }
    auto dma = [&](int k0, int stage) {
        const char* kt_base = reinterpret_cast<const char*>(Kb) + (int64_t)k0 * B_ROWB;
        const char* vt_base = reinterpret_cast<const char*>(Vb) + (int64_t)k0 * B_ROWB;
        char* dst = smem + stage * BQ_TILEB + 1024 * (3 * wave);
        if (k0 + B_T <= Lk) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                __builtin_amdgcn_global_load_lds((b_gptr_t*)(kt_base + g_off[i]), (b_lptr_t*)(dst + 1024 * i), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((b_gptr_t*)(vt_base + g_off[i]), (b_lptr_t*)(dst + B_T * B_ROWB + 1024 * i), 16, 0, 0);
            }
        } else {        // tail tile: rows past Lk re-read the last valid row (finite; their P is masked to 0 below)
            const int last = Lk - 1 - k0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int row = p_row[i] < last ? p_row[i] : last;
                const uint32_t o = (uint32_t)(row * 12 + p_c[i]) * 16u;
                __builtin_amdgcn_global_load_lds((b_gptr_t*)(kt_base + o), (b_lptr_t*)(dst + 1024 * i), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((b_gptr_t*)(vt_base + o), (b_lptr_t*)(dst + B_T * B_ROWB + 1024 * i), 16, 0, 0);
            }
        }
    };
    const int nkt = (Lk + B_T - 1) / B_T;
    dma(0, 0);
    if (nkt > 1) dma(B_T, 1);

    bf16x8 qf[6], dof[6];
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
        if (BWD_ABL & 4) { for (int e = 0; e < 8; ++e) { qf[ks][e] = (short)(lane + e); dof[ks][e] = (short)(lane * 3 + e); } continue; }
        qf[ks] = *reinterpret_cast<const bf16x8*>(Qb + (int64_t)qi * 96 + 16 * ks + 8 * h);
        dof[ks] = *reinterpret_cast<const bf16x8*>(dOrow + 16 * ks + 8 * h);
    }
    float lse = LSE[(int64_t)bh * Lq + qi];
    float dlt;
    if constexpr (FD) {
        const bf16_t* Orow = O + ((int64_t)b * Lq + qi) * C + g * 96;
        bf16x8 of[6];
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) { if (BWD_ABL & 4) { of[ks] = dof[ks]; continue; } of[ks] = *reinterpret_cast<const bf16x8*>(Orow + 16 * ks + 8 * h); }
        float sp = 0.f;
#pragma unroll
        for (int ks = 0; ks < 6; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float o = bf16_to_f32((bf16_t)of[ks][e]);
                if (ADD_Q) o -= bf16_to_f32((bf16_t)qf[ks][e]);          // the forward output carries the pooled-q residual: delta is on the attention part
                sp = fmaf(bf16_to_f32((bf16_t)dof[ks][e]), o, sp);
            }
        dlt = sp + __shfl_xor(sp, 32, 64);                               // the two lanes of a query hold its two half-rows
        if (q_ok && h == 0) delta_out[(int64_t)bh * Lq + qi] = dlt;
    } else {
        dlt = delta[(int64_t)bh * Lq + qi];
    }
    // consume the register operands once: their vmcnt wait is paid here, not inside the loop
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) asm volatile("" : "+v"(qf[ks]), "+v"(dof[ks]));
    // Q is scaled by scale * log2(e) once (and rounded to the 16-bit type again) and every score accumulator starts at -lse: the scores
    // leave the MFMA chain as the exponent itself (P = exp2(S), no multiply-add per element); -delta likewise rides on the dP chain
    if constexpr (PRE) {
        auto scl = [&](short lo, short hi) { return pack_bf16x2(bf16_to_f32((bf16_t)lo) * scale_log2e, bf16_to_f32((bf16_t)hi) * scale_log2e); };
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            const uint4 u = make_uint4(scl(qf[ks][0], qf[ks][1]), scl(qf[ks][2], qf[ks][3]), scl(qf[ks][4], qf[ks][5]), scl(qf[ks][6], qf[ks][7]));
            qf[ks] = *reinterpret_cast<const bf16x8*>(&u);
        }
        if constexpr (FD) {              // ... and leave them for the dK/dV pass, whose query tiles go from memory to LDS without passing through registers
            if (Qs != nullptr && q_ok) {
#pragma unroll
                for (int ks = 0; ks < 6; ++ks) *reinterpret_cast<bf16x8*>(Qs + ((int64_t)bh * Lq + qi) * 96 + 16 * ks + 8 * h) = qf[ks];
            }
        }
    }
    asm volatile("" : "+v"(lse), "+v"(dlt));
    const float ndlt = -dlt, nlse = -lse;
    const float s_init = PRE ? nlse : 0.f;       // !PRE: -lse joins in the exponent's multiply-add instead

    // K / V row fragment of k-step ks: row r, 16-B chunk (2ks + h + rot(r)) mod 12 with rot <= 3: k-steps 0..3 never wrap (immediate
    // offsets from one lane address), k-steps 4 and 5 each get their own
    // transposing reads on the rotation image: row 16*s16 + 4h + (i16>>2) (+8 for the second half), rotation h (+2)
    const int i16 = lane & 15, gi = lane >> 4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    uint32_t ka0, ka4, ka5;
    {
        const int p0 = h + ((r >> 2) & 3);
        const int p4 = p0 + 8 >= 12 ? p0 + 8 - 12 : p0 + 8, p5 = p0 + 10 >= 12 ? p0 + 10 - 12 : p0 + 10;
        ka0 = lds0 + r * B_ROWB + p0 * 16;
        ka4 = lds0 + r * B_ROWB + p4 * 16;
        ka5 = lds0 + r * B_ROWB + p5 * 16;
    }
    uint32_t t_lo[3], t_hi[3];
#pragma unroll
    for (int db = 0; db < 3; ++db) {
        const int c = 4 * db + 2 * (gi & 1) + ((i16 & 3) >> 1);
        int pl = c + h, ph = c + ((h + 2) & 3);
        pl = pl >= 12 ? pl - 12 : pl;
        ph = ph >= 12 ? ph - 12 : ph;
        t_lo[db] = lds0 + (4 * h + (i16 >> 2)) * B_ROWB + 16 * pl + 8 * (i16 & 1);
        t_hi[db] = lds0 + (4 * h + (i16 >> 2) + 8) * B_ROWB + 16 * ph + 8 * (i16 & 1);
    }

    f32x16 dq[3];
#pragma unroll
    for (int db = 0; db < 3; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) dq[db][i] = 0.f;

    int stage = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nkt) dma((kt + 2) * B_T, stage == 0 ? 2 : stage - 1);
        const uint32_t so = (uint32_t)(stage * BQ_TILEB);
        stage = stage == BQ_STAGES - 1 ? 0 : stage + 1;

        // One tile = four phases; inside a wave the dS arithmetic of one 32-key half runs in the gaps between the MFMAs of the
        // other half (MFMA 32 cycles on the matrix pipe, ~8 VALU issues behind each):
        //   A  S^T / dP^T of keys 0..31                          (12 MFMA)
        //   B  S^T / dP^T of keys 32..63  ||  dS of keys 0..31   (12 MFMA, 16 elements)
        //   C  dQ^T += K^T dS^T, keys 0..31  ||  dS of keys 32..63 (6 MFMA, 16 elements)
        //   D  dQ^T += K^T dS^T, keys 32..63                     (6 MFMA)
        // (tile by tile -- 24 MFMA, then all the dS arithmetic, then 12 MFMA -- the matrix pipe idled under 190 VALU instructions
        // and the VALU under 36 MFMAs.)  Keys past Lk (last tile): their score accumulators START at -inf, so P = exp2(-inf) = 0
        // and dS = 0 without a mask in the arithmetic.
        const int kbase = kt * B_T;
        f32x16 s[2], dp[2];
        if (kbase + B_T > Lk) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kbase + 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * h;
                    s[kb][i] = key < Lk ? s_init : -INFINITY;
                    dp[kb][i] = ndlt;
                }
        } else {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) { s[kb][i] = s_init; dp[kb][i] = ndlt; }       // S c - lse and dP^T - delta come out of the MFMA chains
        }
        // K / V row fragments: inline-asm reads two k-steps ahead of the MFMAs with counted waits (LDS returns in order)
        bf16x8 kf[3], vf[3];
        const uint32_t fa0 = ka0 + so, fa4 = ka4 + so, fa5 = ka5 + so;
        float dsv[2][16];
        bf16x8 dsf[4];
        bf16x4 ta[12], tb[12];
#define DQ_RD(SL, A, OFF, KB) { kf[SL] = b_rd128<OFF + (KB) * 32 * B_ROWB>(A); vf[SL] = b_rd128<OFF + B_T * B_ROWB + (KB) * 32 * B_ROWB>(A); }
#define DQ_WAIT(SL, N) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(kf[SL]), "+v"(vf[SL]) : "n"(N))
#define DQ_MM(SL, KS, KB) { s[KB] = mfma16(kf[SL], qf[KS], s[KB]); dp[KB] = mfma16(vf[SL], dof[KS], dp[KB]); }
#define DQ_VAL(KB, I) { const float p_ = __builtin_amdgcn_exp2f(PRE ? s[KB][I] : fmaf(s[KB][I], scale_log2e, nlse)); dsv[KB][I] = p_ * dp[KB][I]; }
#define DQ_SB __builtin_amdgcn_sched_barrier(0);
#define TRQ(A, S16) \
        A[0] = b_tr16<(S16) * 16 * B_ROWB>(t_lo[0] + so); A[1] = b_tr16<(S16) * 16 * B_ROWB>(t_hi[0] + so); \
        A[2] = b_tr16<(S16) * 16 * B_ROWB>(t_lo[1] + so); A[3] = b_tr16<(S16) * 16 * B_ROWB>(t_hi[1] + so); \
        A[4] = b_tr16<(S16) * 16 * B_ROWB>(t_lo[2] + so); A[5] = b_tr16<(S16) * 16 * B_ROWB>(t_hi[2] + so); \
        A[6] = b_tr16<(S16 + 1) * 16 * B_ROWB>(t_lo[0] + so); A[7] = b_tr16<(S16 + 1) * 16 * B_ROWB>(t_hi[0] + so); \
        A[8] = b_tr16<(S16 + 1) * 16 * B_ROWB>(t_lo[1] + so); A[9] = b_tr16<(S16 + 1) * 16 * B_ROWB>(t_hi[1] + so); \
        A[10] = b_tr16<(S16 + 1) * 16 * B_ROWB>(t_lo[2] + so); A[11] = b_tr16<(S16 + 1) * 16 * B_ROWB>(t_hi[2] + so);
        // ---- A
        DQ_RD(0, fa0, 0, 0) DQ_RD(1, fa0, 32, 0)
        DQ_WAIT(0, 2); DQ_RD(2, fa0, 64, 0) DQ_MM(0, 0, 0) DQ_SB
        DQ_WAIT(1, 2); DQ_RD(0, fa0, 96, 0) DQ_MM(1, 1, 0) DQ_SB
        DQ_WAIT(2, 2); DQ_RD(1, fa4, 0, 0) DQ_MM(2, 2, 0) DQ_SB
        DQ_WAIT(0, 2); DQ_RD(2, fa5, 0, 0) DQ_MM(0, 3, 0) DQ_SB
        DQ_WAIT(1, 2); DQ_RD(0, fa0, 0, 1) DQ_MM(1, 4, 0) DQ_SB
        DQ_WAIT(2, 2); DQ_RD(1, fa0, 32, 1) DQ_MM(2, 5, 0) DQ_SB
        // ---- B
        DQ_WAIT(0, 2); DQ_RD(2, fa0, 64, 1) DQ_MM(0, 0, 1) DQ_VAL(0, 0) DQ_VAL(0, 1) DQ_VAL(0, 2) DQ_SB
        DQ_WAIT(1, 2); DQ_RD(0, fa0, 96, 1) DQ_MM(1, 1, 1) DQ_VAL(0, 3) DQ_VAL(0, 4) DQ_VAL(0, 5) DQ_SB
        DQ_WAIT(2, 2); DQ_RD(1, fa4, 0, 1) DQ_MM(2, 2, 1) DQ_VAL(0, 6) DQ_VAL(0, 7) DQ_VAL(0, 8) DQ_SB
        dsf[0] = pack8(&dsv[0][0]);
        DQ_WAIT(0, 2); DQ_RD(2, fa5, 0, 1) DQ_MM(0, 3, 1) DQ_VAL(0, 9) DQ_VAL(0, 10) DQ_VAL(0, 11) DQ_SB
        DQ_WAIT(1, 2);
        TRQ(ta, 0)            // K^T fragments of keys 0..31: in flight under the rest of this phase
        DQ_MM(1, 4, 1) DQ_VAL(0, 12) DQ_VAL(0, 13) DQ_SB
        DQ_WAIT(2, 12); DQ_MM(2, 5, 1) DQ_VAL(0, 14) DQ_VAL(0, 15) DQ_SB
        dsf[1] = pack8(&dsv[0][8]);
        // ---- C
        B_WAIT6(ta, 0);
        TRQ(tb, 2)
#define DQ_DQ(TT, S16, DB, E) dq[DB] = mfma16(b_join(TT[6 * (S16) + 2 * (DB)], TT[6 * (S16) + 2 * (DB) + 1]), dsf[E], dq[DB]);
        DQ_DQ(ta, 0, 0, 0) DQ_VAL(1, 0) DQ_VAL(1, 1) DQ_VAL(1, 2) DQ_SB
        DQ_DQ(ta, 0, 1, 0) DQ_VAL(1, 3) DQ_VAL(1, 4) DQ_VAL(1, 5) DQ_SB
        DQ_DQ(ta, 0, 2, 0) DQ_VAL(1, 6) DQ_VAL(1, 7) DQ_VAL(1, 8) DQ_SB
        dsf[2] = pack8(&dsv[1][0]);
        DQ_DQ(ta, 1, 0, 1) DQ_VAL(1, 9) DQ_VAL(1, 10) DQ_VAL(1, 11) DQ_SB
        DQ_DQ(ta, 1, 1, 1) DQ_VAL(1, 12) DQ_VAL(1, 13) DQ_SB
        DQ_DQ(ta, 1, 2, 1) DQ_VAL(1, 14) DQ_VAL(1, 15) DQ_SB
        dsf[3] = pack8(&dsv[1][8]);
        // ---- D
        B_WAIT6(tb, 0);
        DQ_DQ(tb, 0, 0, 2) DQ_DQ(tb, 0, 1, 2) DQ_DQ(tb, 0, 2, 2)
        DQ_DQ(tb, 1, 0, 3) DQ_DQ(tb, 1, 1, 3) DQ_DQ(tb, 1, 2, 3)
#undef DQ_RD
#undef DQ_WAIT
#undef DQ_MM
#undef DQ_VAL
#undef DQ_SB
#undef DQ_DQ
#undef TRQ
    }
    {   // dQ rows.  Lane (r, h) holds elements 4h .. 4h+3 of every 8-element chunk of its query's row (the accumulator layout) and the
        // whole chunks 2 ks + h of dO (the fragment layout): one v_permlane32_swap pair per k-step turns the dO registers into the
        // halves this lane adds (the + q residual of the forward: dQ += dO; formerly twelve 8-byte reloads per lane), a second pair
        // gives lane (r, 0) the whole chunk 2 ks and lane (r, 1) the whole chunk 2 ks + 1 of the result: six 16-byte stores.
        bf16_t* orow = dQ + ((int64_t)bh * Lq + qi) * 96;
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            // dO halves: swap(word0, word2) -> (first word of my half of chunk 2 ks, of chunk 2 ks + 1); swap(word1, word3) -> second words
            const uint4 dw = *reinterpret_cast<const uint4*>(&dof[ks]);
            const auto s0 = __builtin_amdgcn_permlane32_swap(dw.x, dw.z, false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(dw.y, dw.w, false, false);
            uint32_t outw[2][2];
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const int c = 2 * ks + cc, db = c >> 2, i4 = c & 3;
                float4 v = make_float4(dq[db][4 * i4] * scale, dq[db][4 * i4 + 1] * scale, dq[db][4 * i4 + 2] * scale, dq[db][4 * i4 + 3] * scale);
                if (ADD_Q && !(BWD_ABL & 1)) {
                    const uint32_t w0 = s0[cc], w1 = s1[cc];
                    v.x += lo16_to_f32(w0); v.y += hi16_to_f32(w0); v.z += lo16_to_f32(w1); v.w += hi16_to_f32(w1);
                }
                outw[cc][0] = pack_bf16x2(v.x, v.y);
                outw[cc][1] = pack_bf16x2(v.z, v.w);
            }
            // whole chunks: swap(P[2ks].w, P[2ks+1].w) -> (elements 0..3 half, elements 4..7 half) of the chunk this lane stores
            const auto t0 = __builtin_amdgcn_permlane32_swap(outw[0][0], outw[1][0], false, false);
            const auto t1 = __builtin_amdgcn_permlane32_swap(outw[0][1], outw[1][1], false, false);
            const uint4 o = make_uint4(t0[0], t1[0], t0[1], t1[1]);
            if (q_ok && (!(BWD_ABL & 2) || o.x == 0x12345u)) *reinterpret_cast<uint4*>(orow + 16 * ks + 8 * h) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// pass B (bf16 MFMA): dK, dV.   grid (ceil(Lk/128), B*heads[, splits]), 4 waves x 32 keys; streams 64-query tiles.
// ------------------------------------------------------------------------------------------------
// Same data movement as pass A: the 64-query tiles of Q and dO (one rotation image each, serving the row reads of
// S = Q K^T / dP = dO V^T AND the transposing reads of dV^T += dO^T P / dK^T += Q^T dS) and the tile's lse / delta rows arrive
// through a 3-stage LDS ring filled by global_load_lds (two tiles in flight, one s_barrier per tile, counted vmcnt); nothing
// is staged through registers and nothing is written to LDS by the waves themselves.  (The first version wrote two images
// per operand with ds_write_b128 behind two barriers per tile: the LDS write path alone was ~40 % busy.)
// SPLIT: gridDim.z workgroups share one key block, each sweeping a slice of the query tiles and writing its partial dK / dV
// as an fp32 slab dKf/dVf[z] (plain stores); attn_bwd_dkv_reduce_kernel sums the slabs in z order -- deterministic, no atomics.
#define BK_STAGES 3
#define BK_IMG (B_T * B_ROWB)               // 12 KiB: one 64 x 96 rotation image
#define BK_STAGEB (2 * BK_IMG + 1024)       // Q image | dO image | lse[64] | delta[64] | 2 x 256 B landing pads
#define BK_LDS (BK_STAGES * BK_STAGEB)      // 75 KiB: two workgroups per CU
// PRE (round 5): the query tiles are the pre-scaled 16-bit queries (Qs of the dQ pass / delta kernel), so Q . K^T is the exponent's
// variable part itself.  The wave's K and V fragments are then NEGATED once and the two accumulators of a query block start at + lse / + delta
// of their rows (the float4 reads from the stage land in the accumulator registers): the chains leave lse - S and delta - dP, P = exp2 of the
// negated first (a source modifier) and dS = - P (delta - dP) (another): per 32 x 32 block 16 multiply-adds, 16 subtractions and the
// zero-fill of 32 accumulator registers less.
template <bool SPLIT, bool PRE = false>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kt,
                                                              const bf16_t* __restrict__ V, const bf16_t* __restrict__ dO,
                                                              const float* __restrict__ LSE, const float* __restrict__ delta,
                                                              bf16_t* __restrict__ dK, bf16_t* __restrict__ dV,
                                                              float* __restrict__ dKf, float* __restrict__ dVf, int heads,
                                                              int Lq, int Lk, float scale, float scale_log2e) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // BK_STAGES x BK_STAGEB
    int ktile, bh;
    xcd_group_map(ktile, bh);
    const int b = bh / heads, g = bh - b * heads;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    int ki = ktile * 128 + wave * 32 + r;
    const bool k_ok = ki < Lk;
    ki = k_ok ? ki : Lk - 1;
    const int C = heads * 96;
    const char* Qb = reinterpret_cast<const char*>(Q + (int64_t)bh * Lq * 96);
    const char* dOb = reinterpret_cast<const char*>(dO + (int64_t)b * Lq * C + g * 96);
    const float* Lb = LSE + (int64_t)bh * Lq;
    const float* Db = delta + (int64_t)bh * Lq;

    const int64_t do_ld = (int64_t)C * 2;       // bytes between consecutive dO rows
    const int nqt_all = (Lq + B_T - 1) / B_T;
    const int per_z = (nqt_all + gridDim.z - 1) / gridDim.z;
    const int qt_beg = SPLIT ? blockIdx.z * per_z : 0;
    const int qt_end = SPLIT ? (qt_beg + per_z < nqt_all ? qt_beg + per_z : nqt_all) : nqt_all;
    // DMA pieces (3 per wave and image): LDS position p = 64*piece + lane holds chunk (p%12 - rot(row)) of row p/12.  The piece
    // geometry is recomputed at every call from the lane id (a dozen VALU ops): kept live across the loop it costs six registers
    // the accumulators need
    // The DMA is inline asm (SGPR base + 32-bit lane offset, LDS base in M0): the compiler then does not know that LDS is written
    // asynchronously, so its own LDS reads (row reads and ds_read_b64_tr_b16 builtins, register-coalesced and scheduled by it)
    // carry no vmcnt(0); ordering is the counted s_waitcnt vmcnt + s_barrier at the top of every tile.
    const uint32_t smem_a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    auto dma16 = [&](const char* base, uint32_t off, uint32_t lds) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(off), "s"(base) : "memory");
    };
    auto dma4 = [&](const char* base, uint32_t off, uint32_t lds) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(lds), "v"(off), "s"(base) : "memory");
    };
    auto dma = [&](int qt, int stage) {
        const int q0 = qt * B_T;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(smem_a + stage * BK_STAGEB + 1024 * (3 * wave));
        const int last = Lq - 1 - q0;            // rows past Lq re-read the last valid row (finite; their P is masked to 0 below)
        const char* qbase = Qb + (int64_t)q0 * B_ROWB;           // wave-uniform tile bases
        const char* dbase = dOb + (int64_t)q0 * do_ld;
        int ln = lane;
        asm volatile("" : "+v"(ln));             // piece geometry recomputed per call (a dozen VALU ops) instead of living in registers
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int p = 64 * (3 * wave + i) + ln;
            int row = p / 12;
            const int pos = p - row * 12;
            int c = pos - ((row >> 2) & 3);
            c = c < 0 ? c + 12 : c;
            row = row < last ? row : last;
            dma16(qbase, (uint32_t)(row * 12 + c) * 16u, dst + 1024 * i);
            dma16(dbase, (uint32_t)row * (uint32_t)do_ld + (uint32_t)c * 16u, dst + BK_IMG + 1024 * i);
        }
        // lse (even waves) / delta (odd waves) of the tile's 64 queries: 4 bytes per lane; waves 2, 3 land in the pads so that
        // every wave issues the same number of DMA instructions per tile (one vmcnt count for all)
        const int ql = ln < last ? ln : last;
        dma4(reinterpret_cast<const char*>(((wave & 1) ? Db : Lb) + q0), (uint32_t)ql * 4u,
             __builtin_amdgcn_readfirstlane(smem_a + stage * BK_STAGEB + 2 * BK_IMG + 256 * wave));
    };
    if (qt_beg < qt_end) dma(qt_beg, 0);
    if (qt_beg + 1 < qt_end) dma(qt_beg + 1, 1);

    // The K fragments of the wave's 32 keys stay in registers; the V fragments live in LDS (lane-linear, conflict-free 16-byte
    // reads): with both in registers next to the 96 accumulator registers of dK^T / dV^T the compiler spilled a fragment set to
    // scratch and its reloads' vmcnt(0) drained the DMA ring every block
    bf16x8 kf[6], vf[6];
    {
        const bf16_t* Kb = Kt + (int64_t)bh * Lk * 96;
        const bf16_t* Vb = V + (int64_t)bh * Lk * 96;
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            kf[ks] = *reinterpret_cast<const bf16x8*>(Kb + (int64_t)ki * 96 + 16 * ks + 8 * h);
            vf[ks] = *reinterpret_cast<const bf16x8*>(Vb + (int64_t)ki * 96 + 16 * ks + 8 * h);
        }
        // consume the register operands once: their vmcnt wait is paid here, not inside the loop
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) asm volatile("" : "+v"(kf[ks]), "+v"(vf[ks]));
        if constexpr (PRE) {
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) {
                uint4 a = *reinterpret_cast<const uint4*>(&kf[ks]), c = *reinterpret_cast<const uint4*>(&vf[ks]);
                a.x ^= 0x80008000u; a.y ^= 0x80008000u; a.z ^= 0x80008000u; a.w ^= 0x80008000u;
                c.x ^= 0x80008000u; c.y ^= 0x80008000u; c.z ^= 0x80008000u; c.w ^= 0x80008000u;
                kf[ks] = *reinterpret_cast<const bf16x8*>(&a);
                vf[ks] = *reinterpret_cast<const bf16x8*>(&c);
            }
        }
    }
    int roff[6];
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
        int p = 2 * ks + h + ((r >> 2) & 3);
        p = p >= 12 ? p - 12 : p;
        roff[ks] = p * 16;
    }
    // transposing reads on the rotation image: row 16*s16 + 4h + (i16>>2) (+8 for the second half), rotation h (+2)
    const int i16 = lane & 15, gi = lane >> 4;
    // (chunk position = (4 db + cl + rot) mod 12 with cl, rot <= 3: only db = 2 can wrap, so db = 0 / 1 share one lane address)
    const int cl = 2 * (gi & 1) + ((i16 & 3) >> 1);
    const int rl = cl + h, rh = cl + ((h + 2) & 3);
    const int t_lo0 = (4 * h + (i16 >> 2)) * B_ROWB + 16 * rl + 8 * (i16 & 1);
    const int t_hi0 = (4 * h + (i16 >> 2) + 8) * B_ROWB + 16 * rh + 8 * (i16 & 1);
    const int t_lo2 = (4 * h + (i16 >> 2)) * B_ROWB + 16 * (8 + rl >= 12 ? rl - 4 : 8 + rl) + 8 * (i16 & 1);
    const int t_hi2 = (4 * h + (i16 >> 2) + 8) * B_ROWB + 16 * (8 + rh >= 12 ? rh - 4 : 8 + rh) + 8 * (i16 & 1);
    f32x16 dk[3], dv[3];
#pragma unroll
    for (int db = 0; db < 3; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dk[db][i] = 0.f; dv[db][i] = 0.f; }

    int stage = 0;
    for (int qt = qt_beg; qt < qt_end; ++qt) {
        if (qt + 1 < qt_end) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");      // 7 DMA instructions per wave and tile: this
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // tile has landed, the next may be in flight
        __builtin_amdgcn_s_barrier();                                              // ... for every wave; stage-1's readers are done
        if (qt + 2 < qt_end) dma(qt + 2, stage == 0 ? 2 : stage - 1);
        const char* sQ = smem + stage * BK_STAGEB;
        const float* sL = reinterpret_cast<const float*>(sQ + 2 * BK_IMG);      // lse[64] | delta[64]
        if ((qt + 1) * B_T > Lq) {
            // last, partial tile: its invalid rows were filled from the last valid query (finite data); their lse is set to +inf
            // here, so P = exp2(-inf) = 0 and dS = 0 * finite = 0 -- no masked variant of the block code (a second copy of it in
            // the loop cost 80 registers)
            if (tid < 64 && qt * B_T + tid >= Lq) const_cast<float*>(sL)[tid] = INFINITY;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        stage = stage == BK_STAGES - 1 ? 0 : stage + 1;
        // per 32-query block (one S / dP accumulator pair live: the kernel fits 2 waves per SIMD):
        //   S = Q . K^T and dP = dO . V^T (rows = queries in registers, column = this lane's key), P / dS in registers,
        //   then dV^T += dO^T . P and dK^T += Q^T . dS for the block's two 16-query k-steps
        // LDS reads of the block are issued by hand THREE fragments ahead of the MFMA that consumes them (a ring of four fragment registers,
        // counted lgkmcnt waits).  Left to the compiler every MFMA waited for a read issued one MFMA earlier (register pressure keeps its
        // prefetch distance at one); the second wave of the SIMD covers most of that, so the gain is small: -1.6 % at stage 3, -1.9 % at
        // Lk = 6272, -3 % in blocks 14 / 15, the query-split launches +1 %; train step 46.06 -> 45.84 ms (profiles/r5_attn_dkv_pipelined_reads_ab.txt).
        const uint32_t sQa = smem_a + (uint32_t)(sQ - smem) + r * B_ROWB;          // row reads: + roff[ks] (+ 32 qb rows, + BK_IMG for dO)
        const uint32_t sTa = smem_a + (uint32_t)(sQ - smem);                        // transposing reads: + t_lo0 / t_hi0 / t_lo2 / t_hi2
        auto block = [&](auto qb_tag) {
            constexpr int qb = decltype(qb_tag)::value;
            f32x16 s, dp;
            if constexpr (PRE) {           // rows 32qb + 8g + 4h + 0..3 in registers 4g .. 4g+3
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 l4, d4;
                    if (BWD_ABL & 8) { l4 = make_float4(20.f, 20.f, 20.f, 20.f); d4 = make_float4(0.f, 0.f, 0.f, 0.f); }     // (timing ablation: no lse / delta reads)
                    else {
                        l4 = *reinterpret_cast<const float4*>(sL + 32 * qb + 8 * g + 4 * h);
                        d4 = *reinterpret_cast<const float4*>(sL + 64 + 32 * qb + 8 * g + 4 * h);
                    }
                    s[4 * g] = l4.x; s[4 * g + 1] = l4.y; s[4 * g + 2] = l4.z; s[4 * g + 3] = l4.w;
                    dp[4 * g] = d4.x; dp[4 * g + 1] = d4.y; dp[4 * g + 2] = d4.z; dp[4 * g + 3] = d4.w;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
            }
            bf16x8 fr[4];
            // read i of phase A: k-step i / 2 of the Q image (even i: S chain) or of the dO image (odd i: dP chain)
            auto rdA = [&](auto I) {
                constexpr int i = decltype(I)::value, ks = i / 2;
                if (BWD_ABL & 32) { if (i < 4) fr[i & 3] = kf[i & 3]; return; }          // (timing ablation: no fragment reads)
                fr[i & 3] = b_rd128<32 * qb * B_ROWB + (i & 1) * BK_IMG>(sQa + roff[ks]);
            };
            auto mmA = [&](auto I, auto N) {
                constexpr int i = decltype(I)::value, ks = i / 2;
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fr[i & 3]) : "n"(decltype(N)::value));
                if (BWD_ABL & 64) return;                     // (timing ablation: no MFMAs)
                if constexpr (i & 1) dp = mfma16(fr[i & 3], vf[ks], dp);
                else s = mfma16(fr[i & 3], kf[ks], s);
            };
#define DKV_IC(N) std::integral_constant<int, N>{}
            rdA(DKV_IC(0)); rdA(DKV_IC(1)); rdA(DKV_IC(2));
            rdA(DKV_IC(3)); mmA(DKV_IC(0), DKV_IC(3));
            rdA(DKV_IC(4)); mmA(DKV_IC(1), DKV_IC(3));
            rdA(DKV_IC(5)); mmA(DKV_IC(2), DKV_IC(3));
            rdA(DKV_IC(6)); mmA(DKV_IC(3), DKV_IC(3));
            rdA(DKV_IC(7)); mmA(DKV_IC(4), DKV_IC(3));
            rdA(DKV_IC(8)); mmA(DKV_IC(5), DKV_IC(3));
            rdA(DKV_IC(9)); mmA(DKV_IC(6), DKV_IC(3));
            rdA(DKV_IC(10)); mmA(DKV_IC(7), DKV_IC(3));
            rdA(DKV_IC(11)); mmA(DKV_IC(8), DKV_IC(3));
            mmA(DKV_IC(9), DKV_IC(2));
            mmA(DKV_IC(10), DKV_IC(1));
            mmA(DKV_IC(11), DKV_IC(0));
            // phase B fragment j: group j / 3 = (16-query step sh, product): dV^T += dO^T P (dO image) for sh = 0, then dK^T += Q^T dS (Q image) for
            // sh = 0, then the same for sh = 1; d block db = j % 3.  In this order only P of the first 16 queries has to exist before the first
            // MFMA: dS of those queries, then P and dS of the other 16 are formed under the MFMAs of the groups in front of them.
            auto rdB = [&](auto J) {
                constexpr int j = decltype(J)::value, sh = j / 6, db = j % 3;
                constexpr bool is_k = (j / 3) & 1;
                constexpr int off = (2 * qb + sh) * 16 * B_ROWB + (is_k ? 0 : BK_IMG) + (db == 2 ? 0 : 64 * db);
                if (BWD_ABL & 32) return;
                const bf16x4 lo = b_tr16<off>(sTa + (db == 2 ? t_lo2 : t_lo0));
                const bf16x4 hi = b_tr16<off>(sTa + (db == 2 ? t_hi2 : t_hi0));
                fr[j & 3] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            };
            // the first three transposed fragments do not depend on the exponentials: in flight under them
            rdB(DKV_IC(0)); rdB(DKV_IC(1)); rdB(DKV_IC(2));
            bf16x8 pf[2], dsf[2];
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
                float pv[8], dsv[8];
                if constexpr (PRE && (BWD_ABL & 16)) {        // (timing ablation: no exponentials / products / packing)
                    pf[sh] = vf[sh]; dsf[sh] = kf[sh];
                    asm volatile("" : "+v"(pf[sh]), "+v"(dsf[sh]), "+v"(s), "+v"(dp));
                    continue;
                }
                if constexpr (PRE) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float p = __builtin_amdgcn_exp2f(-s[8 * sh + e]);
                        pv[e] = p;
                        dsv[e] = p * -dp[8 * sh + e];
                    }
                } else
#pragma unroll
                for (int g4 = 0; g4 < 2; ++g4) {
                    // rows 32qb + 8*(2sh+g4) + 4h + 0..3
                    const int q4 = 32 * qb + 8 * (2 * sh + g4) + 4 * h;
                    const float4 l4 = *reinterpret_cast<const float4*>(sL + q4);
                    const float4 d4 = *reinterpret_cast<const float4*>(sL + 64 + q4);
                    const float ls[4] = {l4.x, l4.y, l4.z, l4.w};
                    const float ds4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = 8 * sh + 4 * g4 + e;
                        float p = __builtin_amdgcn_exp2f(fmaf(s[i], scale_log2e, -ls[e]));
                        pv[4 * g4 + e] = p;
                        dsv[4 * g4 + e] = p * (dp[i] - ds4[e]);
                    }
                }
                pf[sh] = pack8(pv);
                dsf[sh] = pack8(dsv);
            }
            auto mmB = [&](auto J, auto N) {       // (each fragment is two reads: N counts reads)
                constexpr int j = decltype(J)::value, sh = j / 6, db = j % 3;
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fr[j & 3]) : "n"(decltype(N)::value));
                if (BWD_ABL & 64) return;
                if constexpr ((j / 3) & 1) dk[db] = mfma16(fr[j & 3], dsf[sh], dk[db]);
                else dv[db] = mfma16(fr[j & 3], pf[sh], dv[db]);
            };
            rdB(DKV_IC(3)); mmB(DKV_IC(0), DKV_IC(6));
            rdB(DKV_IC(4)); mmB(DKV_IC(1), DKV_IC(6));
            rdB(DKV_IC(5)); mmB(DKV_IC(2), DKV_IC(6));
            rdB(DKV_IC(6)); mmB(DKV_IC(3), DKV_IC(6));
            rdB(DKV_IC(7)); mmB(DKV_IC(4), DKV_IC(6));
            rdB(DKV_IC(8)); mmB(DKV_IC(5), DKV_IC(6));
            rdB(DKV_IC(9)); mmB(DKV_IC(6), DKV_IC(6));
            rdB(DKV_IC(10)); mmB(DKV_IC(7), DKV_IC(6));
            rdB(DKV_IC(11)); mmB(DKV_IC(8), DKV_IC(6));
            mmB(DKV_IC(9), DKV_IC(4));
            mmB(DKV_IC(10), DKV_IC(2));
            mmB(DKV_IC(11), DKV_IC(0));
#undef DKV_IC
        };
        block(std::integral_constant<int, 0>{});
        block(std::integral_constant<int, 1>{});
    }
    if (SPLIT) {
        if (k_ok) {
            const int64_t nkv = (int64_t)gridDim.y * Lk * 96;
            float* krow = dKf + (int64_t)blockIdx.z * nkv + ((int64_t)bh * Lk + ki) * 96;
            float* vrow = dVf + (int64_t)blockIdx.z * nkv + ((int64_t)bh * Lk + ki) * 96;
#pragma unroll
            for (int db = 0; db < 3; ++db)
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    const int d = 32 * db + 8 * i4 + 4 * h;
                    store4(krow + d, make_float4(dk[db][4 * i4] * scale, dk[db][4 * i4 + 1] * scale, dk[db][4 * i4 + 2] * scale,
                                                 dk[db][4 * i4 + 3] * scale));
                    store4(vrow + d, make_float4(dv[db][4 * i4], dv[db][4 * i4 + 1], dv[db][4 * i4 + 2], dv[db][4 * i4 + 3]));
                }
        }
        return;
    }
    if (k_ok) {
        bf16_t* krow = dK + ((int64_t)bh * Lk + ki) * 96;
        bf16_t* vrow = dV + ((int64_t)bh * Lk + ki) * 96;
#pragma unroll
        for (int db = 0; db < 3; ++db)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const int d = 32 * db + 8 * i4 + 4 * h;
                store4(krow + d, make_float4(dk[db][4 * i4] * scale, dk[db][4 * i4 + 1] * scale, dk[db][4 * i4 + 2] * scale,
                                             dk[db][4 * i4 + 3] * scale));
                store4(vrow + d, make_float4(dv[db][4 * i4], dv[db][4 * i4 + 1], dv[db][4 * i4 + 2], dv[db][4 * i4 + 3]));
            }
    }
}

// dK / dV = sum over the nz query-slice slabs, in z order (fixed summation order: bit-reproducible), cast to the 16-bit type
__global__ __launch_bounds__(256) void attn_bwd_dkv_reduce_kernel(const float* __restrict__ dKf, const float* __restrict__ dVf,
                                                                  bf16_t* __restrict__ dK, bf16_t* __restrict__ dV, int64_t n4,
                                                                  int nz) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 a = load4(dKf + 4 * i), c = load4(dVf + 4 * i);
        for (int z = 1; z < nz; ++z) {
            const float4 a2 = load4(dKf + 4 * (i + z * n4)), c2 = load4(dVf + 4 * (i + z * n4));
            a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
            c.x += c2.x; c.y += c2.y; c.z += c2.z; c.w += c2.w;
        }
        store4(dK + 4 * i, a);
        store4(dV + 4 * i, c);
    }
}

// ------------------------------------------------------------------------------------------------
// exact fp32 passes (parity path): 4 lanes per row, 24 channels each; streamed tiles of 32 rows in LDS.
// ------------------------------------------------------------------------------------------------
#define F_T 32
__global__ __launch_bounds__(256) void attn_bwd_dq_f32_kernel(const float* __restrict__ Q, const float* __restrict__ Kt,
                                                              const float* __restrict__ V, const float* __restrict__ dO,
                                                              const float* __restrict__ LSE, const float* __restrict__ delta,
                                                              float* __restrict__ dQ, int heads, int Lq, int Lk, float scale,
                                                              int add_q) {
    __shared__ __attribute__((aligned(16))) float sK[F_T * 96];
    __shared__ __attribute__((aligned(16))) float sV[F_T * 96];
    const int bh = blockIdx.y;
    const int b = bh / heads, g = bh - b * heads;
    const int j = threadIdx.x & 3;
    int qi = blockIdx.x * 64 + (threadIdx.x >> 2);
    const bool q_ok = qi < Lq;
    qi = q_ok ? qi : Lq - 1;
    const int C = heads * 96;
    float q[24], d[24], acc[24];
    const float* qrow = Q + ((int64_t)bh * Lq + qi) * 96 + 24 * j;
    const float* drow = dO + ((int64_t)b * Lq + qi) * C + g * 96 + 24 * j;
#pragma unroll
    for (int e = 0; e < 24; ++e) { q[e] = qrow[e]; d[e] = drow[e]; acc[e] = 0.f; }
    const float lse = LSE[(int64_t)bh * Lq + qi] * 0.69314718055994530942f;   // stored in log2 units
    const float dlt = delta[(int64_t)bh * Lq + qi];
    const float* Kb = Kt + (int64_t)bh * Lk * 96;
    const float* Vb = V + (int64_t)bh * Lk * 96;
    for (int k0 = 0; k0 < Lk; k0 += F_T) {
        __syncthreads();
        for (int i = threadIdx.x; i < F_T * 24; i += 256) {
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
        const int nk = min(F_T, Lk - k0);
        for (int kk = 0; kk < nk; ++kk) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int e = 0; e < 24; ++e) {
                s = fmaf(q[e], sK[kk * 96 + 24 * j + e], s);
                dp = fmaf(d[e], sV[kk * 96 + 24 * j + e], dp);
            }
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64);
            dp += __shfl_xor(dp, 1, 64); dp += __shfl_xor(dp, 2, 64);
            const float p = expf(s * scale - lse);
            const float ds = p * (dp - dlt);
#pragma unroll
            for (int e = 0; e < 24; ++e) acc[e] = fmaf(ds, sK[kk * 96 + 24 * j + e], acc[e]);
        }
    }
    if (q_ok) {
        float* o = dQ + ((int64_t)bh * Lq + qi) * 96 + 24 * j;
#pragma unroll
        for (int e = 0; e < 24; ++e) o[e] = acc[e] * scale + (add_q ? d[e] : 0.f);
    }
}

__global__ __launch_bounds__(256) void attn_bwd_dkv_f32_kernel(const float* __restrict__ Q, const float* __restrict__ Kt,
                                                               const float* __restrict__ V, const float* __restrict__ dO,
                                                               const float* __restrict__ LSE, const float* __restrict__ delta,
                                                               float* __restrict__ dK, float* __restrict__ dV, int heads,
                                                               int Lq, int Lk, float scale) {
    __shared__ __attribute__((aligned(16))) float sQ[F_T * 96];
    __shared__ __attribute__((aligned(16))) float sD[F_T * 96];
    __shared__ float sL[2 * F_T];
    const int bh = blockIdx.y;
    const int b = bh / heads, g = bh - b * heads;
    const int j = threadIdx.x & 3;
    int ki = blockIdx.x * 64 + (threadIdx.x >> 2);
    const bool k_ok = ki < Lk;
    ki = k_ok ? ki : Lk - 1;
    const int C = heads * 96;
    float k[24], v[24], ak[24], av[24];
#pragma unroll
    for (int e = 0; e < 24; ++e) {
        k[e] = Kt[((int64_t)bh * Lk + ki) * 96 + 24 * j + e];
        v[e] = V[((int64_t)bh * Lk + ki) * 96 + 24 * j + e];
        ak[e] = 0.f; av[e] = 0.f;
    }
    const float* Qb = Q + (int64_t)bh * Lq * 96;
    const float* dOb = dO + (int64_t)b * Lq * C + g * 96;
    for (int q0 = 0; q0 < Lq; q0 += F_T) {
        __syncthreads();
        for (int i = threadIdx.x; i < F_T * 24; i += 256) {
            const int row = i / 24, c4 = i - row * 24;
            float4 qv = make_float4(0.f, 0.f, 0.f, 0.f), dv4 = qv;
            if (q0 + row < Lq) {
                qv = load4(Qb + (int64_t)(q0 + row) * 96 + 4 * c4);
                dv4 = load4(dOb + (int64_t)(q0 + row) * C + 4 * c4);
            }
            *reinterpret_cast<float4*>(sQ + row * 96 + 4 * c4) = qv;
            *reinterpret_cast<float4*>(sD + row * 96 + 4 * c4) = dv4;
        }
        if (threadIdx.x < 2 * F_T) {
            const int qq = q0 + (threadIdx.x & (F_T - 1));
            sL[threadIdx.x] = qq < Lq ? (threadIdx.x < F_T ? LSE[(int64_t)bh * Lq + qq] * 0.69314718055994530942f
                                                           : delta[(int64_t)bh * Lq + qq])
                                      : (threadIdx.x < F_T ? INFINITY : 0.f);
        }
        __syncthreads();
        const int nq = min(F_T, Lq - q0);
        for (int qq = 0; qq < nq; ++qq) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int e = 0; e < 24; ++e) {
                s = fmaf(k[e], sQ[qq * 96 + 24 * j + e], s);
                dp = fmaf(v[e], sD[qq * 96 + 24 * j + e], dp);
            }
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64);
            dp += __shfl_xor(dp, 1, 64); dp += __shfl_xor(dp, 2, 64);
            const float p = expf(s * scale - sL[qq]);
            const float ds = p * (dp - sL[F_T + qq]);
#pragma unroll
            for (int e = 0; e < 24; ++e) {
                av[e] = fmaf(p, sD[qq * 96 + 24 * j + e], av[e]);
                ak[e] = fmaf(ds, sQ[qq * 96 + 24 * j + e], ak[e]);
            }
        }
    }
    if (k_ok) {
#pragma unroll
        for (int e = 0; e < 24; ++e) {
            dK[((int64_t)bh * Lk + ki) * 96 + 24 * j + e] = ak[e] * scale;
            dV[((int64_t)bh * Lk + ki) * 96 + 24 * j + e] = av[e];
        }
    }
}

static int dkv_splits(int B, int heads, int Lq, int Lk) {
    const int64_t base = (int64_t)B * heads * ((Lk + 127) / 128);
    if (base >= 384) return 1;
    int64_t z = (768 + base - 1) / base;
    const int64_t maxz = (Lq + 1023) / 1024;      // at least 1024 queries per slice
    if (z > maxz) z = maxz;
    return (int)(z < 1 ? 1 : z);
}

bool attn_fwd_prescales_q(int Lq, int Lk);      // attention.hip

// delta [B*heads*Lq] + (split path) fp32 dK, dV partial slabs [2][splits][B*heads*Lk*96] + the pre-scaled 16-bit queries [B*heads*Lq*96]
static int64_t ws_qs_offset_floats(int B, int heads, int Lq, int Lk) {
    const int nz = dkv_splits(B, heads, Lq, Lk);
    const int64_t n = (int64_t)B * heads * Lq + (nz > 1 ? 2ll * nz : 0ll) * B * heads * Lk * 96;
    return (n + 3) & ~3ll;                       // 16-byte aligned
}
extern "C" int64_t mvit_attention_bwd_workspace_bytes(int B, int heads, int Lq, int Lk) {
    return ws_qs_offset_floats(B, heads, Lq, Lk) * (int64_t)sizeof(float) + (int64_t)B * heads * Lq * 96 * 2;
}

// q,k,v as in the forward; out = forward output [B][Lq][heads*96]; lse from the forward; dout same layout as out.
// dq [B][heads][Lq][96], dk/dv [B][heads][Lk][96] (act-typed).  workspace: fp32 [B*heads*Lq] (delta).
extern "C" int mvit_attention_bwd(const void* q, const void* k, const void* v, const void* out, const float* lse,
                                  const void* dout, void* dq, void* dk, void* dv, float* workspace, int B, int heads,
                                  int Lq, int Lk, float scale, int add_q, int act_dtype, void* stream) {
    if (!q || !k || !v || !out || !lse || !dout || !dq || !dk || !dv || !workspace || B <= 0 || heads <= 0 || Lq <= 0 ||
        Lk <= 0)
        return MVIT_EINVAL;
    if ((int64_t)B * heads > 65535) return MVIT_EINVAL;
    hipStream_t st = as_stream(stream);
    const int64_t rows = (int64_t)B * heads * Lq;
    int64_t dblocks = (rows + 63) / 64;
    if (dblocks > 16384) dblocks = 16384;
    const float sl2 = scale * 1.44269504088896340736f;
    if (act_dtype == MVIT_BF16) {
        static const bool flat_delta = getenv("MVIT_ATT_DELTA_ROWS") == nullptr;
        // When the forward ran in the 64-query kernel its scores -- and the saved lse -- are those of round16(q * scale * log2e).  The dQ
        // pass builds the same values in registers; the dK/dV pass streams its query tiles straight into LDS, so the delta kernel
        // (which reads q anyway) leaves it a pre-scaled copy: all three kernels then exponentiate bit-identical scores.  (With the
        // dK/dV pass on the unscaled q, rows with a dominant key -- scores of 20 ... 60 in the exp2 domain -- saw P off by
        // 2^(s * 2^-9): dV errors of 1-2 % instead of 0.3 %, tools/probes/attn_peaked.py.)
        bf16_t* qs = (flat_delta && attn_fwd_prescales_q(Lq, Lk))
                         ? reinterpret_cast<bf16_t*>(workspace + ws_qs_offset_floats(B, heads, Lq, Lk)) : nullptr;
        // Lk <= 2048 (13 of the 16 blocks): dQ and dK/dV run one after the other on this stream, and the dQ pass produces delta and the scaled
        // queries itself (FD form above) -- no delta launch.  The three long-key blocks keep the delta kernel: their two passes run side by side,
        // both need delta at their start.  MVIT_ATT_DELTA_FUSE=0 keeps the separate kernel everywhere (A/B).
        static const bool fd_env = !(getenv("MVIT_ATT_DELTA_FUSE") && getenv("MVIT_ATT_DELTA_FUSE")[0] == '0');
        static const char* side_env0 = getenv("MVIT_ATT_BWD_SIDE");
        const bool fuse_delta = fd_env && flat_delta && !(side_env0 ? side_env0[0] == '1' : Lk > 2048);
        if (fuse_delta) {
        } else if (flat_delta) {
            const int64_t nchunks = rows * 12;
            int64_t fb = (nchunks + 191) / 192;
            if (fb > 8192) fb = 8192;
            hipLaunchKernelGGL(attn_bwd_delta_flat_kernel, dim3((unsigned)fb), dim3(192), 0, st, (const bf16_t*)dout, (const bf16_t*)out,
                               (const bf16_t*)q, workspace, heads, Lq, nchunks, add_q, qs, sl2);
        } else {
            hipLaunchKernelGGL((attn_bwd_delta_kernel<bf16_t>), dim3((unsigned)dblocks), dim3(256), 0, st, (const bf16_t*)dout,
                               (const bf16_t*)out, (const bf16_t*)q, workspace, B, heads, Lq, add_q);
        }
        MVIT_LAUNCH_CHECK();
        // the dQ pass (on st) and the dK/dV pass only share their inputs and delta: fork here, right behind the delta kernel, and
        // issue the dK/dV pass on the library's side stream so the two passes fill each other's partial last waves of workgroups
        // Measured (profiles/r2_attn_bwd_side_ab.txt): side by side the two passes are 2-4 % faster than one after the other when
        // the key loop is long (Lk = 6272, the three transition blocks) and 7-11 % SLOWER for Lk = 1568 (each pass alone already fills
        // the chip; interleaved workgroups of two different kernels share the CUs badly).  MVIT_ATT_BWD_SIDE=0/1 forces.
        static const char* side_env = getenv("MVIT_ATT_BWD_SIDE");
        const bool want_side = side_env ? side_env[0] == '1' : Lk > 2048;
        SideStream* ss = want_side ? side_stream_for_current_device() : nullptr;
        hipStream_t skv = (ss && side_fork(ss, st)) ? ss->side : st;
        // (a 64-queries-per-wave form of pass A was built in round 3 and measured 4-12 % behind this kernel, profiles/r3_attn_dq_w64.txt:
        // it lives in tools/probes/attention_bwd_w64.hip, outside the library, since round 5)
        dim3 gq((Lq + 127) / 128, B * heads);
        static DevFlags dq_attr_done_tab; DevFlag dq_attr_done = dev_flag(dq_attr_done_tab);
        if (!dq_attr_done) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, BQ_STAGES * BQ_TILEB) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, BQ_STAGES * BQ_TILEB) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, BQ_STAGES * BQ_TILEB) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, BQ_STAGES * BQ_TILEB) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, BQ_STAGES * BQ_TILEB) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, BQ_STAGES * BQ_TILEB) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, BQ_STAGES * BQ_TILEB) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, BQ_STAGES * BQ_TILEB) != hipSuccess)
                return MVIT_ELAUNCH;
            dq_attr_done = true;
        }
        // the dQ pass exponentiates the scores the forward built its lse on: pre-scaled 16-bit queries when that ran in the 64-query
        // kernel, (q . k) * scale * log2e otherwise (short sequences, MVIT_ATT_W64=0)
        const bool pre = attn_fwd_prescales_q(Lq, Lk);
#define DQ_LAUNCH(AQ, PR) hipLaunchKernelGGL((attn_bwd_dq_kernel<AQ, PR>), gq, dim3(256), BQ_STAGES * BQ_TILEB, st, (const bf16_t*)q, (const bf16_t*)k, \
                               (const bf16_t*)v, (const bf16_t*)dout, lse, workspace, (bf16_t*)dq, heads, Lq, Lk, scale, sl2)
#define DQ_LAUNCH_FD(AQ, PR) hipLaunchKernelGGL((attn_bwd_dq_kernel<AQ, PR, true>), gq, dim3(256), BQ_STAGES * BQ_TILEB, st, (const bf16_t*)q, (const bf16_t*)k, \
                               (const bf16_t*)v, (const bf16_t*)dout, lse, workspace, (bf16_t*)dq, heads, Lq, Lk, scale, sl2, (const bf16_t*)out, workspace, qs)
        if (fuse_delta) {
            if (add_q) { if (pre) DQ_LAUNCH_FD(true, true); else DQ_LAUNCH_FD(true, false); }
            else { if (pre) DQ_LAUNCH_FD(false, true); else DQ_LAUNCH_FD(false, false); }
        } else if (add_q) { if (pre) DQ_LAUNCH(true, true); else DQ_LAUNCH(true, false); }
        else { if (pre) DQ_LAUNCH(false, true); else DQ_LAUNCH(false, false); }
#undef DQ_LAUNCH
#undef DQ_LAUNCH_FD
        MVIT_LAUNCH_CHECK();
        static DevFlags dkv_attr_done_tab; DevFlag dkv_attr_done = dev_flag(dkv_attr_done_tab);
        if (!dkv_attr_done) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, BK_LDS) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, BK_LDS) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, BK_LDS) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, BK_LDS) != hipSuccess)
                return MVIT_ELAUNCH;
            dkv_attr_done = true;
        }
        const int nz = dkv_splits(B, heads, Lq, Lk);
        static const bool dkv_pre = !(getenv("MVIT_ATT_DKV_PRE") && atoi(getenv("MVIT_ATT_DKV_PRE")) == 0);      // A/B switch
        // pre-scaled queries: the scores come out of the MFMA in the exp2 domain (multiplier 1), and dK = scale * dS^T q =
        // (scale / (scale log2e)) * dS^T qs
        const bf16_t* q_kv = qs ? qs : (const bf16_t*)q;
        const float sl2_kv = qs ? 1.0f : sl2, scale_kv = qs ? scale / sl2 : scale;
        const bool pre_kv = qs && dkv_pre;
        float* dkf = nz > 1 ? workspace + rows : nullptr;
        const int64_t nkv = (int64_t)B * heads * Lk * 96;
        float* dvf = nz > 1 ? dkf + (int64_t)nz * nkv : nullptr;
        dim3 gk((Lk + 127) / 128, B * heads, nz > 1 ? nz : 1);
#define DKV_LAUNCH(SP, PR) hipLaunchKernelGGL((attn_bwd_dkv_kernel<SP, PR>), gk, dim3(256), BK_LDS, skv, q_kv, (const bf16_t*)k, (const bf16_t*)v, \
                                          (const bf16_t*)dout, lse, workspace, (bf16_t*)dk, (bf16_t*)dv, dkf, dvf, heads, Lq, Lk, scale_kv, sl2_kv)
        if (nz > 1) { if (pre_kv) DKV_LAUNCH(true, true); else DKV_LAUNCH(true, false); }
        else { if (pre_kv) DKV_LAUNCH(false, true); else DKV_LAUNCH(false, false); }
#undef DKV_LAUNCH
        MVIT_LAUNCH_CHECK();
        if (nz > 1) {
            int64_t cb = (nkv / 4 + 255) / 256;
            if (cb > 4096) cb = 4096;
            hipLaunchKernelGGL(attn_bwd_dkv_reduce_kernel, dim3((unsigned)cb), dim3(256), 0, skv, dkf, dvf, (bf16_t*)dk, (bf16_t*)dv, nkv / 4, nz);
            MVIT_LAUNCH_CHECK();
        }
        if (skv != st && !side_join(ss, st)) return MVIT_ELAUNCH;
        return MVIT_OK;
    }
    if (act_dtype != MVIT_F32) return MVIT_EDTYPE;
    hipLaunchKernelGGL((attn_bwd_delta_kernel<float>), dim3((unsigned)dblocks), dim3(256), 0, st, (const float*)dout,
                       (const float*)out, (const float*)q, workspace, B, heads, Lq, add_q);
    MVIT_LAUNCH_CHECK();
    dim3 gq((Lq + 63) / 64, B * heads);
    hipLaunchKernelGGL(attn_bwd_dq_f32_kernel, gq, dim3(256), 0, st, (const float*)q, (const float*)k, (const float*)v,
                       (const float*)dout, lse, workspace, (float*)dq, heads, Lq, Lk, scale, add_q);
    MVIT_LAUNCH_CHECK();
    dim3 gk((Lk + 63) / 64, B * heads);
    hipLaunchKernelGGL(attn_bwd_dkv_f32_kernel, gk, dim3(256), 0, st, (const float*)q, (const float*)k, (const float*)v,
                       (const float*)dout, lse, workspace, (float*)dk, (float*)dv, heads, Lq, Lk, scale);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}
