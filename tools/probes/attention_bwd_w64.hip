// Attention backward, pass A (dQ), head_dim 96: 64 queries per wave, ONE wave per SIMD -- the form of attention_w64.hip applied to
// attn_bwd_dq_kernel (attention_bwd.hip; reference math: autograd of slowfast/models/attention.py:267-279).
//   dQ[q] = scale * sum_k dS[q][k] K[k] (+ dO[q] for the pooled-q residual),  dS = P (dP - delta),  P = exp2(S c - LSE2),  dP = dO V^T
// A wave owns two 32-query blocks: every K / V row fragment read from LDS feeds two MFMAs of S^T = K Q^T and two of dP^T = V dO^T,
// every K^T fragment two of dQ^T += K^T dS^T; a K/V tile is streamed once per 256 queries.
//   ACC registers (named in the asm text, never seen by the compiler):
//     a[0:95]    dQ^T, six 32x32 tiles (query block j, 32-d block db: tile 3j + db)
//     a[96:143]  Q^T fragments (4 (6j + ks)), a[144:191] dO^T fragments (4 (6j + ks)), written once
//     a[192:239] K^T fragments of the tile (16-key step s, d block db: 4 (3s + db)), ds_read_b64_tr_b16 pairs
//   arch VGPRs: S^T and dP^T of the tile (2 x 64), the packed dS fragments (32), three rotating K / V row fragments, statistics.
// One key tile = 72 MFMAs in four phases (as in the 32-query kernel, every count doubled):
//   A  S^T / dP^T of keys 0..31 (24)            B  S^T / dP^T of keys 32..63 (24)  beside  dS of keys 0..31
//   C  dQ^T += K^T dS^T, keys 0..31 (12)  beside  dS of keys 32..63            D  dQ^T += K^T dS^T, keys 32..63 (12)
// The dS arithmetic of an element is cut into three ops (fma + sub / exp / mul + pack) and software-pipelined over the element
// stream, so no op follows the op it depends on (one wave per SIMD: nothing else hides a dependency stall); every op is pinned
// to its MFMA slot by an empty asm naming its results.  The ragged key tile is visited first (a sum does not care about order):
// its masked keys start their score accumulators at -inf in straight-line code of its own, the loop never masks.
#include <type_traits>

#include "common.h"

#define Y_T 64
#define Y_ROWB 192
#define Y_TILEB (2 * Y_T * Y_ROWB)      // K image | V image: 24 KiB
#define Y_STAGES 3
#define Y_QB 256
#define YA_Q 96
#define YA_DO 144
#define YA_KT 192
#define Y_CLOB_ALL "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239"
#ifdef MVIT_HALF_IS_FP16
#define Y_MFMA "v_mfma_f32_32x32x16_f16 "
#else
#define Y_MFMA "v_mfma_f32_32x32x16_bf16 "
#endif
#define Y_SB __builtin_amdgcn_sched_barrier(0)
template <int N> using YC = std::integral_constant<int, N>;
typedef __attribute__((address_space(1))) const void y_gptr_t;
typedef __attribute__((address_space(3))) void y_lptr_t;

template <int BR>
__device__ __forceinline__ void y_mm(f32x16& acc, const bf16x8& a) {      // acc += a (rows of K / V) . B fragment in a[BR:BR+3]
    asm volatile(Y_MFMA "%0, %1, a[%c2:%c3], %0" : "+v"(acc) : "v"(a), "i"(BR), "i"(BR + 3));
}
template <int BR>
__device__ __forceinline__ void y_mm0(f32x16& acc, const bf16x8& a) {     // acc = a . B fragment (first k-step: C = 0)
    asm volatile(Y_MFMA "%0, %1, a[%c2:%c3], 0" : "=&v"(acc) : "v"(a), "i"(BR), "i"(BR + 3));      // early clobber: D must not share registers with A
}
template <int BR>
__device__ __forceinline__ void y_mmc(f32x16& acc, const bf16x8& a, const f32x16& c) {     // acc = a . B fragment + c (first k-step of dP^T: c = -delta)
    asm volatile(Y_MFMA "%0, %1, a[%c3:%c4], %2" : "=&v"(acc) : "v"(a), "v"(c), "i"(BR), "i"(BR + 3));
}
template <int OR_, int AR>
__device__ __forceinline__ void y_dq(const bf16x8& ds) {                  // dQ^T tile += K^T fragment (A, ACC) . dS^T fragment (B)
    asm volatile(Y_MFMA "a[%c1:%c2], a[%c3:%c4], %0, a[%c1:%c2]" ::"v"(ds), "i"(OR_), "i"(OR_ + 15), "i"(AR), "i"(AR + 3));
}
template <int R, int OFF>
__device__ __forceinline__ void y_trd(uint32_t addr) {
    asm volatile("ds_read_b64_tr_b16 a[%c1:%c2], %0 offset:%c3" ::"v"(addr), "i"(R), "i"(R + 1), "i"(OFF));
}
template <int OFF>
__device__ __forceinline__ bf16x8 y_rd128(uint32_t addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int R>
__device__ __forceinline__ void y_put(const uint4& u) {
    asm volatile("v_accvgpr_write_b32 a%c4, %0\n\tv_accvgpr_write_b32 a%c5, %1\n\tv_accvgpr_write_b32 a%c6, %2\n\tv_accvgpr_write_b32 a%c7, %3"
                 ::"v"(u.x), "v"(u.y), "v"(u.z), "v"(u.w), "i"(R), "i"(R + 1), "i"(R + 2), "i"(R + 3));
}
template <int R>
__device__ __forceinline__ void y_get4(float4& v) {
    asm volatile("v_accvgpr_read_b32 %0, a%c4\n\tv_accvgpr_read_b32 %1, a%c5\n\tv_accvgpr_read_b32 %2, a%c6\n\tv_accvgpr_read_b32 %3, a%c7"
                 : "=v"(v.x), "=v"(v.y), "=v"(v.z), "=v"(v.w) : "i"(R), "i"(R + 1), "i"(R + 2), "i"(R + 3));
}
template <int I, int N, typename F>
__device__ __forceinline__ void y_for(F&& f) {
    if constexpr (I < N) {
        f(YC<I>{});
        y_for<I + 1, N>(f);
    }
}

template <bool ADD_Q>
__global__ __launch_bounds__(256, 1) void attn_bwd_dq_w64_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kt,
                                                                 const bf16_t* __restrict__ V, const bf16_t* __restrict__ dO,
                                                                 const float* __restrict__ LSE, const float* __restrict__ delta,
                                                                 bf16_t* __restrict__ dQ, int heads, int Lq, int Lk, float scale,
                                                                 float scale_log2e) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // Y_STAGES x (K rotation image | V rotation image)
    int qtile, bh;
    xcd_group_map(qtile, bh);
    const int b = bh / heads, g = bh - b * heads;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int C = heads * 96;
    const bf16_t* Qb = Q + (int64_t)bh * Lq * 96;
    const bf16_t* Kb = Kt + (int64_t)bh * Lk * 96;
    const bf16_t* Vb = V + (int64_t)bh * Lk * 96;

    asm volatile("v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_write_b32 a1, 0\n\tv_accvgpr_write_b32 a2, 0\n\tv_accvgpr_write_b32 a3, 0\n\tv_accvgpr_write_b32 a4, 0\n\tv_accvgpr_write_b32 a5, 0\n\tv_accvgpr_write_b32 a6, 0\n\tv_accvgpr_write_b32 a7, 0\n\tv_accvgpr_write_b32 a8, 0\n\tv_accvgpr_write_b32 a9, 0\n\tv_accvgpr_write_b32 a10, 0\n\tv_accvgpr_write_b32 a11, 0\n\tv_accvgpr_write_b32 a12, 0\n\tv_accvgpr_write_b32 a13, 0\n\tv_accvgpr_write_b32 a14, 0\n\tv_accvgpr_write_b32 a15, 0\n\tv_accvgpr_write_b32 a16, 0\n\tv_accvgpr_write_b32 a17, 0\n\tv_accvgpr_write_b32 a18, 0\n\tv_accvgpr_write_b32 a19, 0\n\tv_accvgpr_write_b32 a20, 0\n\tv_accvgpr_write_b32 a21, 0\n\tv_accvgpr_write_b32 a22, 0\n\tv_accvgpr_write_b32 a23, 0\n\tv_accvgpr_write_b32 a24, 0\n\tv_accvgpr_write_b32 a25, 0\n\tv_accvgpr_write_b32 a26, 0\n\tv_accvgpr_write_b32 a27, 0\n\tv_accvgpr_write_b32 a28, 0\n\tv_accvgpr_write_b32 a29, 0\n\tv_accvgpr_write_b32 a30, 0\n\tv_accvgpr_write_b32 a31, 0\n\tv_accvgpr_write_b32 a32, 0\n\tv_accvgpr_write_b32 a33, 0\n\tv_accvgpr_write_b32 a34, 0\n\tv_accvgpr_write_b32 a35, 0\n\tv_accvgpr_write_b32 a36, 0\n\tv_accvgpr_write_b32 a37, 0\n\tv_accvgpr_write_b32 a38, 0\n\tv_accvgpr_write_b32 a39, 0\n\tv_accvgpr_write_b32 a40, 0\n\tv_accvgpr_write_b32 a41, 0\n\tv_accvgpr_write_b32 a42, 0\n\tv_accvgpr_write_b32 a43, 0\n\tv_accvgpr_write_b32 a44, 0\n\tv_accvgpr_write_b32 a45, 0\n\tv_accvgpr_write_b32 a46, 0\n\tv_accvgpr_write_b32 a47, 0\n\tv_accvgpr_write_b32 a48, 0\n\tv_accvgpr_write_b32 a49, 0\n\tv_accvgpr_write_b32 a50, 0\n\tv_accvgpr_write_b32 a51, 0\n\tv_accvgpr_write_b32 a52, 0\n\tv_accvgpr_write_b32 a53, 0\n\tv_accvgpr_write_b32 a54, 0\n\tv_accvgpr_write_b32 a55, 0\n\tv_accvgpr_write_b32 a56, 0\n\tv_accvgpr_write_b32 a57, 0\n\tv_accvgpr_write_b32 a58, 0\n\tv_accvgpr_write_b32 a59, 0\n\tv_accvgpr_write_b32 a60, 0\n\tv_accvgpr_write_b32 a61, 0\n\tv_accvgpr_write_b32 a62, 0\n\tv_accvgpr_write_b32 a63, 0\n\tv_accvgpr_write_b32 a64, 0\n\tv_accvgpr_write_b32 a65, 0\n\tv_accvgpr_write_b32 a66, 0\n\tv_accvgpr_write_b32 a67, 0\n\tv_accvgpr_write_b32 a68, 0\n\tv_accvgpr_write_b32 a69, 0\n\tv_accvgpr_write_b32 a70, 0\n\tv_accvgpr_write_b32 a71, 0\n\tv_accvgpr_write_b32 a72, 0\n\tv_accvgpr_write_b32 a73, 0\n\tv_accvgpr_write_b32 a74, 0\n\tv_accvgpr_write_b32 a75, 0\n\tv_accvgpr_write_b32 a76, 0\n\tv_accvgpr_write_b32 a77, 0\n\tv_accvgpr_write_b32 a78, 0\n\tv_accvgpr_write_b32 a79, 0\n\tv_accvgpr_write_b32 a80, 0\n\tv_accvgpr_write_b32 a81, 0\n\tv_accvgpr_write_b32 a82, 0\n\tv_accvgpr_write_b32 a83, 0\n\tv_accvgpr_write_b32 a84, 0\n\tv_accvgpr_write_b32 a85, 0\n\tv_accvgpr_write_b32 a86, 0\n\tv_accvgpr_write_b32 a87, 0\n\tv_accvgpr_write_b32 a88, 0\n\tv_accvgpr_write_b32 a89, 0\n\tv_accvgpr_write_b32 a90, 0\n\tv_accvgpr_write_b32 a91, 0\n\tv_accvgpr_write_b32 a92, 0\n\tv_accvgpr_write_b32 a93, 0\n\tv_accvgpr_write_b32 a94, 0\n\tv_accvgpr_write_b32 a95, 0\n\t" ::: Y_CLOB_ALL);      // dQ^T = 0; the clobber list is what reserves a[0:239] for the asm text
    int qi[2];
    bool q_ok[2];
    float nlse[2], dlt[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        qi[j] = qtile * Y_QB + wave * 64 + 32 * j + r;
        q_ok[j] = qi[j] < Lq;
        qi[j] = q_ok[j] ? qi[j] : Lq - 1;
        nlse[j] = -LSE[(int64_t)bh * Lq + qi[j]];
        dlt[j] = delta[(int64_t)bh * Lq + qi[j]];
    }
    {   // all 24 fragment loads in flight together, then the ACC writes (load / wait / write per fragment cost ~12 serial HBM round trips)
        uint4 uq[12], ud[12];
#pragma unroll
        for (int I = 0; I < 12; ++I) {
            const int j = I / 6, ks = I % 6;
            uq[I] = *reinterpret_cast<const uint4*>(Qb + (int64_t)qi[j] * 96 + 16 * ks + 8 * h);
            ud[I] = *reinterpret_cast<const uint4*>(dO + ((int64_t)b * Lq + qi[j]) * C + g * 96 + 16 * ks + 8 * h);
        }
        y_for<0, 12>([&](auto I) {
            y_put<YA_Q + 4 * I>(uq[I]);
            y_put<YA_DO + 4 * I>(ud[I]);
        });
    }

    // LDS-DMA pieces: LDS position p = 64*piece + lane holds chunk (p%12 - rot(row)) of row p/12 (K and V alike); 3 + 3 per wave and tile
    const int nkt = (Lk + Y_T - 1) / Y_T;
    const bool ragged = (Lk % Y_T) != 0;
#ifdef Y_RAGGED_LAST
    auto tile_key0 = [&](int tile) { return tile * Y_T; };
#else
    auto tile_key0 = [&](int tile) { return ragged ? (tile == 0 ? (nkt - 1) * Y_T : (tile - 1) * Y_T) : tile * Y_T; };
#endif
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
    auto dma = [&](int tile, int stage) {
        const int k0 = tile_key0(tile);
        const char* kt_base = reinterpret_cast<const char*>(Kb) + (int64_t)k0 * Y_ROWB;
        const char* vt_base = reinterpret_cast<const char*>(Vb) + (int64_t)k0 * Y_ROWB;
        char* dst = smem + stage * Y_TILEB + 1024 * (3 * wave);
        if (k0 + Y_T <= Lk) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                __builtin_amdgcn_global_load_lds((y_gptr_t*)(kt_base + g_off[i]), (y_lptr_t*)(dst + 1024 * i), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((y_gptr_t*)(vt_base + g_off[i]), (y_lptr_t*)(dst + Y_T * Y_ROWB + 1024 * i), 16, 0, 0);
            }
        } else {        // ragged tile: rows past Lk re-read the last valid row (finite; their P is exp2(-inf) = 0)
            const int last = Lk - 1 - k0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int row = p_row[i] < last ? p_row[i] : last;
                const uint32_t o = (uint32_t)(row * 12 + p_c[i]) * 16u;
                __builtin_amdgcn_global_load_lds((y_gptr_t*)(kt_base + o), (y_lptr_t*)(dst + 1024 * i), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((y_gptr_t*)(vt_base + o), (y_lptr_t*)(dst + Y_T * Y_ROWB + 1024 * i), 16, 0, 0);
            }
        }
    };
    dma(0, 0);
    if (nkt > 1) dma(1, 1);

    // row fragments: row r (+32 kb), 16-B chunk (2ks + h + rot(r)) mod 12; transposing reads: row 16 s + 4h + (i16>>2) (+8), rotation h (+2)
    const int i16 = lane & 15, gi = lane >> 4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    uint32_t ka0, ka4, ka5;
    {
        const int p0 = h + ((r >> 2) & 3);
        const int p4 = p0 + 8 >= 12 ? p0 + 8 - 12 : p0 + 8, p5 = p0 + 10 >= 12 ? p0 + 10 - 12 : p0 + 10;
        ka0 = lds0 + r * Y_ROWB + p0 * 16;
        ka4 = lds0 + r * Y_ROWB + p4 * 16;
        ka5 = lds0 + r * Y_ROWB + p5 * 16;
    }
    uint32_t t_lo[3], t_hi[3];
#pragma unroll
    for (int db = 0; db < 3; ++db) {
        const int c = 4 * db + 2 * (gi & 1) + ((i16 & 3) >> 1);
        int pl = c + h, ph = c + ((h + 2) & 3);
        pl = pl >= 12 ? pl - 12 : pl;
        ph = ph >= 12 ? ph - 12 : ph;
        t_lo[db] = lds0 + (4 * h + (i16 >> 2)) * Y_ROWB + 16 * pl + 8 * (i16 & 1);
        t_hi[db] = lds0 + (4 * h + (i16 >> 2) + 8) * Y_ROWB + 16 * ph + 8 * (i16 & 1);
    }

#ifdef Y_STAMP
    uint64_t tacc[5] = {0, 0, 0, 0, 0}, tprev = 0;
#define Y_T0() { Y_SB; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev) :: "memory"); Y_SB; }
#define Y_TS(N) { uint64_t tn_; Y_SB; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tn_) :: "memory"); Y_SB; tacc[N] += tn_ - tprev; tprev = tn_; }
#else
#define Y_T0()
#define Y_TS(N)
#endif
    // ---- one key tile ------------------------------------------------------------------------------------------------------
    f32x16 s[2][2], dp[2][2];           // [query block][32-key block]
    bf16x8 dsf[2][4];                   // packed dS^T fragments [query block][16-key step]
    uint32_t pw[2][4];
    float e_t[3], e_p[3], e_hold[2];
    f32x16 ndl[2];                      // -delta of the lane's query in all 16 rows: the initial value of every dP^T accumulator
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int i = 0; i < 16; ++i) ndl[j][i] = -dlt[j];
    }
    asm volatile("s_nop 4" : "+v"(ndl[0]), "+v"(ndl[1]));
    // dS arithmetic of key block KB, element stream v = 16 j + i (34 steps: 32 elements + two draining the pipeline)
    auto val = [&](auto KB, auto Vv) {
        constexpr int kb = KB, v = Vv;
        if constexpr (v >= 1 && v <= 32) {                    // exp of element v - 1
            constexpr int k = (v - 1) % 3;
            float& p_ = e_p[k];
            p_ = __builtin_amdgcn_exp2f(e_t[k]);
            asm volatile("" : "+v"(p_));
        }
        if constexpr (v < 32) {                               // fma of element v (dP^T - delta comes out of the MFMA chain)
            constexpr int j = v / 16, i = v % 16, k = v % 3;
            float& t_ = e_t[k];
            t_ = __builtin_fmaf(s[j][kb][i], scale_log2e, nlse[j]);
            asm volatile("" : "+v"(t_));
        }
        if constexpr (v >= 2) {                               // mul + pack of element v - 2
            constexpr int w = v - 2, j = w / 16, i = w % 16, k = w % 3;
            const float dsv = e_p[k] * dp[j][kb][i];
            if constexpr ((i & 1) == 0) {
                float& h_ = e_hold[j];
                h_ = dsv;
                asm volatile("" : "+v"(h_));
            } else {
                uint32_t& w_ = pw[j][(i % 8) / 2];
                w_ = pack_bf16x2(e_hold[j], dsv);
                asm volatile("" : "+v"(w_));
                if constexpr ((i % 8) == 7) {
                    const uint4 u = make_uint4(pw[j][0], pw[j][1], pw[j][2], pw[j][3]);
                    bf16x8& f_ = dsf[j][2 * kb + i / 8];
                    f_ = *reinterpret_cast<const bf16x8*>(&u);
                    asm volatile("" : "+v"(f_));          // the fragment exists from here on (a slot ahead of the MFMA that reads it)
                }
            }
        }
    };
    auto iter = [&](int i, uint32_t so, int k0, auto first_tag, auto mask_tag) {
        constexpr bool FIRST = decltype(first_tag)::value, MASK = decltype(mask_tag)::value;
        bf16x8 kf[3], vf[3];
        const uint32_t fa0 = ka0 + so, fa4 = ka4 + so, fa5 = ka5 + so;
        if constexpr (MASK) {           // the ragged tile: masked keys start at -inf (P = 0, dS = 0 with no mask in the arithmetic)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int key = k0 + 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * h;
                        s[j][kb][i] = key < Lk ? 0.f : -INFINITY;
                    }
            // materialised HERE: left to the compiler the register copies land directly in front of the asm MFMA that takes the tile as
            // C, and nothing pads a VALU write in front of an asm MFMA's operand read (seen: garbage dQ on every ragged shape)
            asm volatile("s_nop 4" : "+v"(s[0][0]), "+v"(s[0][1]), "+v"(s[1][0]), "+v"(s[1][1]));
        }
        // fragment read of k-step KS, key block KB into slot SL (K row fragment | V row fragment)
        auto rd = [&](auto SL, auto KS, auto KB) {
            constexpr int sl = SL, ks = KS, kb = KB;
            constexpr int off = (ks < 4 ? 32 * ks : 0) + kb * 32 * Y_ROWB;
            const uint32_t a = ks < 4 ? fa0 : (ks == 4 ? fa4 : fa5);
            kf[sl] = y_rd128<off>(a);
            vf[sl] = y_rd128<off + Y_T * Y_ROWB>(a);
        };
        auto wait = [&](auto SL, auto N) {
            constexpr int sl = SL, n = N;
            bf16x8 &ka_ = kf[sl], &va_ = vf[sl];          // (asm operands cannot name a captured array element directly)
            asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(ka_), "+v"(va_) : "n"(n));
        };
        // the four MFMAs of k-step KS, key block KB: MFMA number M of them (0, 1: S^T of query block 0, 1; 2, 3: dP^T)
        auto mm = [&](auto SL, auto KS, auto KB, auto M) {
            constexpr int sl = SL, ks = KS, kb = KB, m = M, j = m & 1;
            if constexpr (m < 2) {
                if constexpr (ks == 0 && !MASK) y_mm0<YA_Q + 4 * (6 * j + ks)>(s[j][kb], kf[sl]);
                else y_mm<YA_Q + 4 * (6 * j + ks)>(s[j][kb], kf[sl]);
            } else {
                if constexpr (ks == 0) y_mmc<YA_DO + 4 * (6 * j + ks)>(dp[j][kb], vf[sl], ndl[j]);       // dP^T - delta comes out of the chain
                else y_mm<YA_DO + 4 * (6 * j + ks)>(dp[j][kb], vf[sl]);
            }
        };
        // K^T fragments of 16-key steps S0, S0 + 1 (12 transposing reads each step)
        auto trq = [&](auto S0) {
            y_for<0, 6>([&](auto I) {
                constexpr int sidx = S0 + I / 3, db = I % 3;
                constexpr int R = YA_KT + 4 * (3 * sidx + db);
                y_trd<R, sidx * 16 * Y_ROWB>(t_lo[db] + so);
                y_trd<R + 2, sidx * 16 * Y_ROWB>(t_hi[db] + so);
            });
        };
        // this wave's six LDS-DMA pieces of tile i+2 (clamped past the end: the piece lands in a stage nobody reads any more), one per slot
        const int dt = i + 2 < nkt ? i + 2 : nkt - 1;
        const char* d_k = reinterpret_cast<const char*>(Kb) + (int64_t)tile_key0(dt) * Y_ROWB;
        const char* d_v = reinterpret_cast<const char*>(Vb) + (int64_t)tile_key0(dt) * Y_ROWB;
        char* d_dst = smem + ((i + 2) % Y_STAGES) * Y_TILEB + 1024 * (3 * wave);
        auto dma_piece = [&](auto P) {
            constexpr int pc_ = P, pi = pc_ / 2;
            if constexpr ((pc_ & 1) == 0) __builtin_amdgcn_global_load_lds((y_gptr_t*)(d_k + g_off[pi]), (y_lptr_t*)(d_dst + 1024 * pi), 16, 0, 0);
            else __builtin_amdgcn_global_load_lds((y_gptr_t*)(d_v + g_off[pi]), (y_lptr_t*)(d_dst + Y_T * Y_ROWB + 1024 * pi), 16, 0, 0);
        };
        Y_TS(0)
        // ---- A(i): S^T / dP^T of keys 0..31 beside dS of keys 32..63 of tile i-1 (34 stream steps over 24 MFMA slots)
        rd(YC<0>{}, YC<0>{}, YC<0>{});
        rd(YC<1>{}, YC<1>{}, YC<0>{});
        y_for<0, 6>([&](auto KS) {
            constexpr int ks = KS, sl = ks % 3;
            wait(YC<sl>{}, YC<2>{});
            // two k-steps ahead: k-steps 2..5 of this key block, then 0, 1 of the next
            if constexpr (ks + 2 < 6) rd(YC<(ks + 2) % 3>{}, YC<ks + 2>{}, YC<0>{});
            else rd(YC<(ks + 2) % 3>{}, YC<ks + 2 - 6>{}, YC<1>{});
            y_for<0, 4>([&](auto M) {
                constexpr int slot = 4 * ks + M;
                mm(YC<sl>{}, KS, YC<0>{}, M);
                if constexpr (!FIRST) y_for<(slot * 34) / 24, ((slot + 1) * 34) / 24>([&](auto Vv) { val(YC<1>{}, Vv); });
                Y_SB;
            });
        });
        Y_TS(1)
        // ---- D(i-1): dQ^T += K^T dS^T over keys 32..63 of tile i-1 (its K^T fragments were read in C(i-1))
        if constexpr (!FIRST) {
            y_for<0, 12>([&](auto I) {
                constexpr int sidx = 2 + I / 6, db = (I / 2) % 3, j = I & 1;
                y_dq<16 * (3 * j + db), YA_KT + 4 * (3 * sidx + db)>(dsf[j][sidx]);
                Y_SB;
            });
        }
        Y_TS(2)
        // ---- B(i): S^T / dP^T of keys 32..63 beside dS of keys 0..31
        y_for<0, 6>([&](auto KS) {
            constexpr int ks = KS, sl = ks % 3;
            if constexpr (ks < 4) {
                wait(YC<sl>{}, YC<2>{});
                rd(YC<(ks + 2) % 3>{}, YC<ks + 2>{}, YC<1>{});
            } else if constexpr (ks == 4) {
                wait(YC<sl>{}, YC<2>{});
                trq(YC<0>{});            // K^T fragments of keys 0..31: in flight under the rest of this phase
            } else {
                wait(YC<sl>{}, YC<12>{});
            }
            y_for<0, 4>([&](auto M) {
                constexpr int slot = 4 * ks + M;
                mm(YC<sl>{}, KS, YC<1>{}, M);
                y_for<(slot * 34) / 24, ((slot + 1) * 34) / 24>([&](auto Vv) { val(YC<0>{}, Vv); });
                Y_SB;
            });
        });
        Y_TS(3)
        // ---- C(i): dQ^T += K^T dS^T over keys 0..31; the K^T fragments of keys 32..63 are requested for D(i), the DMA pieces go out
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        trq(YC<2>{});
        y_for<0, 12>([&](auto I) {
            constexpr int sidx = I / 6, db = (I / 2) % 3, j = I & 1;
            y_dq<16 * (3 * j + db), YA_KT + 4 * (3 * sidx + db)>(dsf[j][sidx]);
            if constexpr (I < 6) dma_piece(I);
            Y_SB;
        });
        Y_TS(4)
    };

    // iteration i: A(i) | D(i-1) | B(i) | C(i); the first one is straight-line code of its own (no tile before it; the ragged tile)
    {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");       // (a single tile: both prologue groups are its own, vmcnt(0) below covers it)
        if (nkt == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (ragged) iter(0, 0u, (nkt - 1) * Y_T, std::true_type{}, std::true_type{});
        else iter(0, 0u, 0, std::true_type{}, std::false_type{});
    }
    for (int i = 1; i < nkt; ++i) {
        Y_T0()
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        iter(i, (uint32_t)((i % Y_STAGES) * Y_TILEB), 0, std::false_type{}, std::false_type{});
    }
    // drain: dS of keys 32..63 of the last tile, then its D
    y_for<0, 34>([&](auto Vv) { val(YC<1>{}, Vv); });
    Y_SB;
    y_for<0, 12>([&](auto I) {
        constexpr int sidx = 2 + I / 6, db = (I / 2) % 3, j = I & 1;
        y_dq<16 * (3 * j + db), YA_KT + 4 * (3 * sidx + db)>(dsf[j][sidx]);
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the clamped DMA pieces of the last iterations land before the workgroup gives its LDS back

#ifdef Y_STAMP
    if (lane == 0) {       // diagnostic build: cycles per tile in (wait + barrier + DMA issue, A, B, C, D) land in the first rows of dQ (as bf16-rounded floats / 16)
        float* dbg = reinterpret_cast<float*>(dQ + ((int64_t)bh * Lq + qtile * Y_QB + wave * 64) * 96);
        for (int i = 0; i < 5; ++i) dbg[i] = (float)tacc[i] / nkt;
    }
    return;
#endif
    // ---- dQ rows: scale, + dO for the pooled-q residual, store [bh][q][96] ----------------------------------------------------
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    y_for<0, 2>([&](auto J) {
        constexpr int j = J;
        bf16_t* orow = dQ + ((int64_t)bh * Lq + qi[j]) * 96;
        const bf16_t* dOrow = dO + ((int64_t)b * Lq + qi[j]) * C + g * 96;
        y_for<0, 12>([&](auto I) {
            constexpr int db = I / 4, i4 = I % 4;
            float4 v;
            y_get4<16 * (3 * j + db) + 4 * i4>(v);
            const int d = 32 * db + 8 * i4 + 4 * h;
            v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
            if (ADD_Q) {
                const float4 dd = load4(dOrow + d);
                v.x += dd.x; v.y += dd.y; v.z += dd.z; v.w += dd.w;
            }
            if (q_ok[j]) store4(orow + d, v);
        });
    });
}

int attn_bwd_dq_w64_prepare() {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_w64_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, Y_STAGES * Y_TILEB) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_w64_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, Y_STAGES * Y_TILEB) != hipSuccess)
        return MVIT_ELAUNCH;
    return MVIT_OK;
}

// launcher used by mvit_attention_bwd (attention_bwd.hip) when this form is selected
int attn_bwd_dq_w64_launch(const void* q, const void* k, const void* v, const void* dout, const float* lse, const float* delta, void* dq, int B,
                           int heads, int Lq, int Lk, float scale, float scale_log2e, int add_q, hipStream_t st) {
    dim3 grid((Lq + Y_QB - 1) / Y_QB, B * heads);
    if (add_q)
        hipLaunchKernelGGL((attn_bwd_dq_w64_kernel<true>), grid, dim3(256), Y_STAGES * Y_TILEB, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,
                           (const bf16_t*)dout, lse, delta, (bf16_t*)dq, heads, Lq, Lk, scale, scale_log2e);
    else
        hipLaunchKernelGGL((attn_bwd_dq_w64_kernel<false>), grid, dim3(256), Y_STAGES * Y_TILEB, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,
                           (const bf16_t*)dout, lse, delta, (bf16_t*)dq, heads, Lq, Lk, scale, scale_log2e);
    return MVIT_OK;
}
