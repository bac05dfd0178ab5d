// Fused pooled attention forward, head_dim 96, 64 queries per wave, ONE wave per SIMD, software-pipelined by hand
// (reference math: slowfast/models/attention.py:267-279).  The DEFAULT forward for Lq >= 128, Lk >= 64 in the 16-bit builds (attention.hip routes here;
// MVIT_ATT_W64=0 selects the 32-query kernels of attention.hip for A/B runs; short sequences always use those).
//
// Why this shape: in the 32-query kernels every K / V fragment read from LDS feeds one MFMA and a K/V tile is streamed once per
// 128 queries; their timing ablations show the fragment reads, the LDS-DMA and the MFMAs costing their own
// time one after the other.  Here a wave owns two 32-query blocks, every fragment feeds two MFMAs, a tile is streamed once per
// 256 queries, and the wave has the whole 512-register file:
//   ACC registers (named in the asm text, never seen by the compiler):
//     a[0:95]    O^T, six 32x32 tiles (query block j, 32-d block db: tile 3j + db)
//     a[96:143]  Q^T fragments (query block j, k-step ks: 4 (6j + ks)), written once
//     a[144:191] K fragments of the tile whose scores are computed next (k-step ks, 32-key block kb: 4 (2ks + kb)), ds_read_b128
//     a[192:239] V^T fragments of the tile being accumulated (16-key step s, 32-d block db: 4 (3s + db)), ds_read_b64_tr_b16 pairs
//   arch VGPRs (compiler-allocated): the score tiles of two key tiles (2 x 64), their packed probabilities (2 x 32), statistics.
// One key tile = two phases of 24 MFMAs:
//   phase 1:  S(t+1) = K(t+1) Q^T        beside  V^T(t) fragment reads, the second half of tile t's exponentials
//   phase 2:  O^T += V^T(t) P(t)^T       beside  K(t+2) fragment reads, the row maxima of S(t+1), the first half of its exponentials
// one s_barrier per tile; K tiles arrive by LDS-DMA three tiles ahead, V tiles one ahead, two buffers each.
// Around the key loop (round 5; ablation builds showed a quarter of the kernel in an item's first loads and last stores):
// a wave's 64 query rows arrive by LDS-DMA as whole rows in its own LDS region and stay there for the + q residual; the finished 16-bit output
// tile goes back into the same slots and leaves as whole rows; the grid is persistent (at most one workgroup per CU, items v = blockIdx.x,
// + gridDim.x, ...) and a workgroup requests its NEXT item's Q tile and first K / V tiles before the epilogue of the current one (two regions per wave).
// Q is multiplied by scale * log2(e) once (and rounded to the 16-bit type again) and every score accumulator STARTS at minus the row's
// reference point, so the scores leave the MFMA chain as the exponent itself: a pair of scores costs two v_exp, two adds and a pack.
// Every MFMA is an asm statement followed by its share of the softmax arithmetic and a scheduling barrier, so the instruction
// stream is the one written here.  Hazards the compiler cannot see (it pads nothing around asm): an S tile written by asm MFMAs is
// first read a whole phase later; a P fragment written by compiler VALU is read by an asm MFMA at least one slot later.
#include "common.h"

#ifndef W_ABL
#define W_ABL 0        // timing ablations (tools/r5_w64_abl.sh; results invalid): 1 exp -> mul, 2 no exponential work, 4 no row maxima, 8 no fragment reads,
#endif                 // 16 no LDS-DMA in the loop, 32 no QK MFMAs, 64 no PV MFMAs, 128 no epilogue loads / stores, 256 no Q loads
#define W_KT 64
#define W_ROWB 192
#define W_TILE (W_KT * W_ROWB)           // 12 KiB
#define W_QB 256
#define W_SMEM (12 * W_TILE)             // K0 | K1 | V0 | V1 | two Q / output tiles per wave (this item's, the next / previous one's)
#define W_LAG 8.0f                     // log2 of the largest probability value a lagging row maximum may produce
#define WA_Q 96
#define WA_K 144
#define WA_V 192
#define W_CLOB_ALL "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239"
#ifdef MVIT_HALF_IS_FP16
#define W_MFMA "v_mfma_f32_32x32x16_f16 "
#else
#define W_MFMA "v_mfma_f32_32x32x16_bf16 "
#endif
typedef __attribute__((ext_vector_type(2))) float wf32x2;

template <int KR, int QR>
__device__ __forceinline__ void w_qk(f32x16& s) {      // s += K frag (A) . Q^T frag (B)
    asm volatile(W_MFMA "%0, a[%c1:%c2], a[%c3:%c4], %0" : "+v"(s) : "i"(KR), "i"(KR + 3), "i"(QR), "i"(QR + 3));
}
template <int KR, int QR>
__device__ __forceinline__ void w_qk0(f32x16& s) {     // s = K frag . Q^T frag (first k-step of a chain: C = 0)
    asm volatile(W_MFMA "%0, a[%c1:%c2], a[%c3:%c4], 0" : "=v"(s) : "i"(KR), "i"(KR + 3), "i"(QR), "i"(QR + 3));
}
template <int KR, int QR>
__device__ __forceinline__ void w_qkc(f32x16& s, const f32x16& c) {      // s = K frag . Q^T frag + c (first k-step: c = -reference point)
    asm volatile(W_MFMA "%0, a[%c2:%c3], a[%c4:%c5], %1" : "=&v"(s) : "v"(c), "i"(KR), "i"(KR + 3), "i"(QR), "i"(QR + 3));
}
template <int OR_, int VR>
__device__ __forceinline__ void w_pv(const bf16x8& p) {      // O^T tile += V^T frag (A) . P^T frag (B)
    asm volatile(W_MFMA "a[%c1:%c2], a[%c3:%c4], %0, a[%c1:%c2]" ::"v"(p), "i"(OR_), "i"(OR_ + 15), "i"(VR), "i"(VR + 3));
}
template <int R, int OFF>
__device__ __forceinline__ void w_krd(uint32_t addr) {
    asm volatile("ds_read_b128 a[%c1:%c2], %0 offset:%c3" ::"v"(addr), "i"(R), "i"(R + 3), "i"(OFF));
}
template <int R, int OFF>
__device__ __forceinline__ void w_vrd(uint32_t addr) {
    asm volatile("ds_read_b64_tr_b16 a[%c1:%c2], %0 offset:%c3" ::"v"(addr), "i"(R), "i"(R + 1), "i"(OFF));
}
template <int R>
__device__ __forceinline__ void w_qput(const uint4& u) {
    asm volatile("v_accvgpr_write_b32 a%c4, %0\n\tv_accvgpr_write_b32 a%c5, %1\n\tv_accvgpr_write_b32 a%c6, %2\n\tv_accvgpr_write_b32 a%c7, %3"
                 ::"v"(u.x), "v"(u.y), "v"(u.z), "v"(u.w), "i"(R), "i"(R + 1), "i"(R + 2), "i"(R + 3));
}
template <int R>
__device__ __forceinline__ void w_oget4(float4& v) {
    asm volatile("v_accvgpr_read_b32 %0, a%c4\n\tv_accvgpr_read_b32 %1, a%c5\n\tv_accvgpr_read_b32 %2, a%c6\n\tv_accvgpr_read_b32 %3, a%c7"
                 : "=v"(v.x), "=v"(v.y), "=v"(v.z), "=v"(v.w) : "i"(R), "i"(R + 1), "i"(R + 2), "i"(R + 3));
}
template <int R>
__device__ __forceinline__ void w_oscale(float alpha) {     // a[R] *= alpha
    float t;
    asm volatile("v_accvgpr_read_b32 %0, a%c2\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a%c2, %0" : "=&v"(t) : "v"(alpha), "i"(R));
}
template <int I, int N, typename F>
__device__ __forceinline__ void w_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        w_for<I + 1, N>(f);
    }
}
#define W_SB __builtin_amdgcn_sched_barrier(0)
template <int N> using IC = std::integral_constant<int, N>;

// (the last two kernel arguments are unused: they belonged to the key-split form of the ragged query tile, tools/probes/attn_fwd_keysplit.patch)
template <bool ADD_Q>
__global__ __launch_bounds__(256, 1) void attn_fwd_w64_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kt,
                                                              const bf16_t* __restrict__ V, bf16_t* __restrict__ O,
                                                              float* __restrict__ LSE, int heads, int Lq, int Lk_all, float scale_log2e,
                                                              float* __restrict__ /*unused*/, int n_bh) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef W_STAMP
    float* const LSE_stamp = LSE;      // diagnostic build: the LSE buffer carries the stamps instead (averages over a workgroup's items, in
    LSE = nullptr;                     // the slot of its last one)
    float sacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int n_it = 0;
#endif
    const int Lk = Lk_all;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const bool is_v = wave >= 2;
    // Work items = (256-query tile, batch x head) pairs, taken v = blockIdx.x, + gridDim.x, ... by a grid of at most one workgroup per CU
    // (attn_fwd_w64_launch): a workgroup that has a next item requests its Q tile and first K / V tiles BEFORE the epilogue of the
    // current one, so that memory latency runs under the epilogue instead of in front of the next prologue, and the finished output
    // tile leaves LDS only after the next item's operands have landed.  Item -> (tile, group): xcd_group_map's rule on the virtual
    // index (the grid is a multiple of 8 whenever a workgroup takes more than one item, so v keeps its workgroup's XCD).
    const int nqt = (Lq + W_QB - 1) / W_QB, n_items = nqt * n_bh;
    auto item_of = [&](int v, int& qt, int& grp) {
        if ((n_bh & 7) == 0) {
            const int xcd = v & 7, slot = v >> 3, gq = slot / nqt;
            grp = gq * 8 + xcd;
            qt = slot - gq * nqt;
        } else {
            grp = v / nqt;
            qt = v - grp * nqt;
        }
    };
    int v_item = blockIdx.x, qtile, bh;
    item_of(v_item, qtile, bh);
    int q0 = qtile * W_QB + wave * 64;
    const bf16_t* Qb = Q + (int64_t)bh * Lq * 96;
    // LDS-DMA: waves 0, 1 move K tiles, waves 2, 3 V tiles, six 1-KiB pieces each (layout and swizzle as in attention.hip)
    const char* src_bh = reinterpret_cast<const char*>((is_v ? V : Kt) + (int64_t)bh * Lk_all * 96);
    auto piece_off = [&](int i, int ln, int last_row) -> uint32_t {
        const int p = 64 * (6 * (wave & 1) + i) + ln;
        int row = p / 12, c = p - row * 12;
        if (!is_v) {
            c -= (row >> 2) & 3;
            c = c < 0 ? c + 12 : c;
        }
        row = row < last_row ? row : last_row;
        return (uint32_t)(row * 12 + c) * 16u;
    };
    uint32_t g_off[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) g_off[i] = piece_off(i, lane, W_KT);
    const uint32_t smem_a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)(smem);
    const uint32_t ring_a = smem_a + (is_v ? 2 * W_TILE : 0) + 1024 * (6 * (wave & 1));
    auto dma1 = [&](const char* base, uint32_t off, uint32_t lds) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(off), "s"(base) : "memory");
    };
    // Key tiles are visited ragged-tail-first: the online softmax does not care about the order, and the one tile that needs its
    // keys masked is then the prologue's (straight-line code of its own) -- the pipelined loop never masks.
    const int nkt = (Lk + W_KT - 1) / W_KT;
    const bool ragged = (Lk % W_KT) != 0;
    auto tile_key0 = [&](int tile) { return ragged ? (tile == 0 ? (nkt - 1) * W_KT : (tile - 1) * W_KT) : tile * W_KT; };
    auto dma = [&](const char* kv_bh, int tile) {               // this wave's six pieces of K / V tile `tile` into buffer tile & 1
        const int k0 = tile_key0(tile);
        const char* t_base = kv_bh + (int64_t)k0 * W_ROWB;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(ring_a + (tile & 1) * W_TILE);
        if (k0 + W_KT <= Lk) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                dma1(t_base, g_off[i], dst + 1024 * i);
                dma1(t_base + 16 * W_ROWB, g_off[i], dst + 1024 * (i + 3));
            }
        } else {        // tail tile: rows past Lk re-read the last valid row (finite data; their scores are masked to -inf)
            int ln = lane;
            asm volatile("" : "+v"(ln));
#pragma unroll
            for (int i = 0; i < 6; ++i) dma1(t_base, piece_off(i, ln, Lk - 1 - k0), dst + 1024 * i);
        }
    };
    // fragment addresses (tile buffer 0): K k-steps 0..3 from one lane address + immediates, k-steps 4, 5 wrap the rotation
    uint32_t ka0, ka4, ka5;
    {
        const int p0 = h + ((r >> 2) & 3);
        const int p4 = p0 + 8 >= 12 ? p0 + 8 - 12 : p0 + 8, p5 = p0 + 10 >= 12 ? p0 + 10 - 12 : p0 + 10;
        ka0 = smem_a + r * W_ROWB + p0 * 16;
        ka4 = smem_a + r * W_ROWB + p4 * 16;
        ka5 = smem_a + r * W_ROWB + p5 * 16;
    }
    const int i16 = lane & 15, gi = lane >> 4;
    const uint32_t va0 = smem_a + 2 * W_TILE + (4 * h + (i16 >> 2)) * W_ROWB + (16 * (gi & 1) + 4 * (i16 & 3)) * 2;

    // the K fragment reads of one tile (12 ds_read_b128), I = 2 ks + kb; and the V^T reads (12 pairs), I = 3 s + db
    auto k_read = [&](auto I, uint32_t kb_off) {
        constexpr int ks = I / 2, kb = I % 2;
        constexpr int R = WA_K + 4 * I;
        if constexpr (ks < 4) w_krd<R, 32 * ks + 32 * W_ROWB * kb>(ka0 + kb_off);
        else if constexpr (ks == 4) w_krd<R, 32 * W_ROWB * kb>(ka4 + kb_off);
        else w_krd<R, 32 * W_ROWB * kb>(ka5 + kb_off);
    };
    auto v_read = [&](auto I, uint32_t vb_off) {
        constexpr int s = I / 3, db = I % 3;
        constexpr int R = WA_V + 4 * I;
        w_vrd<R, s * 16 * W_ROWB + db * 64>(va0 + vb_off);
        w_vrd<R + 2, s * 16 * W_ROWB + db * 64 + 8 * W_ROWB>(va0 + vb_off);
    };
    // QK MFMA number I of a tile: k-step I / 4, accumulator (j, kb) = ((I / 2) & 1, I & 1)
    f32x16 nref[2];                      // -(reference point) of the lane's query in all 16 rows: where every score accumulator starts
    auto qk = [&](auto I, f32x16 (&s)[2][2], auto first_tag) {
        constexpr int ks = I / 4, j = (I / 2) & 1, kb = I & 1;
        if constexpr (ks == 0 && decltype(first_tag)::value) w_qk0<WA_K + 4 * (2 * ks + kb), WA_Q + 4 * (6 * j + ks)>(s[j][kb]);
        else if constexpr (ks == 0) w_qkc<WA_K + 4 * (2 * ks + kb), WA_Q + 4 * (6 * j + ks)>(s[j][kb], nref[j]);
        else w_qk<WA_K + 4 * (2 * ks + kb), WA_Q + 4 * (6 * j + ks)>(s[j][kb]);
    };
    // PV MFMA number I: 16-key step I / 6, d block (I / 2) % 3, query block I & 1
    auto pv = [&](auto I, bf16x8 (&pf)[2][4]) {
        constexpr int s = I / 6, db = (I / 2) % 3, j = I & 1;
        w_pv<16 * (3 * j + db), WA_V + 4 * (3 * s + db)>(pf[j][s]);
    };

    float m_run[2], l_run[2] = {0.f, 0.f};
    // Exponentials of one pair of scores -- S tile (j, kb = u / 2), registers 8 (u % 2) + 2 jj, + 1 -> word jj of P fragment u of
    // query block j -- cut into three stages so that the work of 16 pairs spreads evenly over 24 MFMA slots.  Every stage ends
    // pinned (an empty asm naming its results): without a use the optimiser sinks the arithmetic out of its slot.
    uint32_t pw[2][4][4];
    wf32x2 e_p[2];                     // in-flight pair state (two pairs are in flight in the pipelined phases)
    auto ex_b0 = [&](f32x16 (&s)[2][2], auto J, auto E, auto K) {
        constexpr int j = J, e = E, k = K, u = e / 4, jj = e % 4, kb = u / 2, sh = u % 2;
        if constexpr (W_ABL & 2) return;
        e_p[k][0] = (W_ABL & 1) ? s[j][kb][8 * sh + 2 * jj] * 1.0001f : __builtin_amdgcn_exp2f(s[j][kb][8 * sh + 2 * jj]);
        asm volatile("" : "+v"(e_p[k][0]));
    };
    auto ex_b1 = [&](f32x16 (&s)[2][2], auto J, auto E, auto K) {
        constexpr int j = J, e = E, k = K, u = e / 4, jj = e % 4, kb = u / 2, sh = u % 2;
        if constexpr (W_ABL & 2) return;
        e_p[k][1] = (W_ABL & 1) ? s[j][kb][8 * sh + 2 * jj + 1] * 1.0001f : __builtin_amdgcn_exp2f(s[j][kb][8 * sh + 2 * jj + 1]);
        asm volatile("" : "+v"(e_p[k][1]));
    };
    auto ex_c = [&](bf16x8 (&pf)[2][4], auto J, auto E, auto K, wf32x2& ps) {             // row sum, pack; fourth word closes the fragment
        constexpr int j = J, e = E, k = K, u = e / 4, jj = e % 4;
        if constexpr (W_ABL & 2) return;
        ps[0] += e_p[k][0];
        ps[1] += e_p[k][1];
        pw[j][u][jj] = pack_bf16x2(e_p[k][0], e_p[k][1]);
        asm volatile("" : "+v"(pw[j][u][jj]), "+v"(ps[0]), "+v"(ps[1]));
        if constexpr (jj == 3) {
            const uint4 v = make_uint4(pw[j][u][0], pw[j][u][1], pw[j][u][2], pw[j][u][3]);
            pf[j][u] = *reinterpret_cast<const bf16x8*>(&v);
            asm volatile("" : "+v"(pf[j][u]));        // the fragment exists from here on: no register copy lands in front of the asm MFMA reading it
        }
    };
    // pairs E0 .. E1-1 of query block J, one after the other (prologue, last tile)
    auto ex_range = [&](f32x16 (&s)[2][2], bf16x8 (&pf)[2][4], auto J, auto E0, auto E1, wf32x2& ps) {
        w_for<decltype(E0)::value, decltype(E1)::value>([&](auto E) { ex_b0(s, J, E, IC<0>{}); ex_b1(s, J, E, IC<0>{}); ex_c(pf, J, E, IC<0>{}, ps); });
    };
    // row maxima of a score tile pair (both 32-key blocks, both wave halves), keys >= Lk of the ragged last tile masked first
    auto tile_max = [&](f32x16 (&s)[2][2], float (&mx)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float m = fmaxf(s[j][0][0], s[j][1][0]);
#pragma unroll
            for (int i = 1; i < 16; ++i) m = fmaxf(fmaxf(m, s[j][0][i]), s[j][1][i]);
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
            mx[j] = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
    };

    // ---- prologue: K(0), K(1), V(0) and this wave's Q tile on their way together; S(0), its softmax; K(1) fragments; K(2) ------
    // The wave's 64 query rows go through LDS (its own 12-KiB region behind the K / V ring, K's row image): whole 192-byte rows by
    // LDS-DMA instead of twelve 32-byte-per-row register loads, one memory latency for Q, K and V together, and the raw rows stay
    // there for the epilogue's + q and as the staging area of the output tile (stored as whole rows as well).  Measured with the
    // ablation builds (profiles/r5_attn_w64_ablations.txt): the register loads and the per-lane 8-byte loads / stores of the epilogue
    // were 24 % of the kernel at Lk = 1568.
    auto qo_region = [&](int p) -> uint32_t { return __builtin_amdgcn_readfirstlane(smem_a + (4 + 2 * wave + p) * W_TILE); };
    auto qo_slot = [&](int ln, int i, int& row, int& chunk) {      // 16-byte slot 64 i + lane of a region holds chunk `chunk` of tile row `row`
        const int p = 64 * i + ln;
        row = p / 12;
        chunk = p - row * 12 - ((row >> 2) & 3);
        chunk = chunk < 0 ? chunk + 12 : chunk;
    };
    auto issue_loads = [&](const char* kv_bh, const bf16_t* q_bh, int q0_, uint32_t qo) {      // an item's first K / V tiles and Q tile
        dma(kv_bh, 0);
        if (!is_v && nkt > 1) dma(kv_bh, 1);
        int ln = lane;                  // (opaque: the slot arithmetic is recomputed where it is used, not kept in registers across the key loop)
        asm volatile("" : "+v"(ln));
        if (!(W_ABL & 256)) {
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                int row, chunk;
                qo_slot(ln, i, row, chunk);
                row = q0_ + row < Lq ? q0_ + row : Lq - 1;          // rows past Lq re-read the last one (never stored)
                dma1(reinterpret_cast<const char*>(q_bh), (uint32_t)(row * W_ROWB + chunk * 16), qo + 1024 * i);
            }
        }
    };
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) u32x2* lds_u2p;
    typedef const __attribute__((address_space(3))) u32x4* lds_u4p;
    auto store_rows = [&](uint32_t qo, char* obase, int q0_) {       // a finished 64 x 96 tile: LDS region -> whole rows of the output
        const int C = heads * 96;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        u32x4 u[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) u[i] = *(lds_u4p)(uintptr_t)(qo + 1024 * i + 16 * ln);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            int row, chunk;
            qo_slot(ln, i, row, chunk);
            if (q0_ + row < Lq && !(W_ABL & 128)) *reinterpret_cast<u32x4*>(obase + (int64_t)row * C * 2 + chunk * 16) = u[i];
        }
    };
    int par = 0;
    char* pend_obase = nullptr;       // the previous item's output tile still waits in region par ^ 1
    int pend_q0 = 0;
    issue_loads(src_bh, Qb, q0, qo_region(0));
  for (;;) {
    const uint32_t qo_a = qo_region(par);
    l_run[1] = 0.f;
#ifdef W_STAMP
    uint64_t t_entry;
    W_SB; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_entry) :: "memory"); W_SB;
#endif
    int Lk_it = Lk;                   // opaque per item: the 64 lane masks of the ragged tile are not worth 128 scalar registers across the loop
    asm volatile("" : "+s"(Lk_it));
    asm volatile("v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_write_b32 a1, 0\n\tv_accvgpr_write_b32 a2, 0\n\tv_accvgpr_write_b32 a3, 0\n\tv_accvgpr_write_b32 a4, 0\n\tv_accvgpr_write_b32 a5, 0\n\tv_accvgpr_write_b32 a6, 0\n\tv_accvgpr_write_b32 a7, 0\n\tv_accvgpr_write_b32 a8, 0\n\tv_accvgpr_write_b32 a9, 0\n\tv_accvgpr_write_b32 a10, 0\n\tv_accvgpr_write_b32 a11, 0\n\tv_accvgpr_write_b32 a12, 0\n\tv_accvgpr_write_b32 a13, 0\n\tv_accvgpr_write_b32 a14, 0\n\tv_accvgpr_write_b32 a15, 0\n\tv_accvgpr_write_b32 a16, 0\n\tv_accvgpr_write_b32 a17, 0\n\tv_accvgpr_write_b32 a18, 0\n\tv_accvgpr_write_b32 a19, 0\n\tv_accvgpr_write_b32 a20, 0\n\tv_accvgpr_write_b32 a21, 0\n\tv_accvgpr_write_b32 a22, 0\n\tv_accvgpr_write_b32 a23, 0\n\tv_accvgpr_write_b32 a24, 0\n\tv_accvgpr_write_b32 a25, 0\n\tv_accvgpr_write_b32 a26, 0\n\tv_accvgpr_write_b32 a27, 0\n\tv_accvgpr_write_b32 a28, 0\n\tv_accvgpr_write_b32 a29, 0\n\tv_accvgpr_write_b32 a30, 0\n\tv_accvgpr_write_b32 a31, 0\n\tv_accvgpr_write_b32 a32, 0\n\tv_accvgpr_write_b32 a33, 0\n\tv_accvgpr_write_b32 a34, 0\n\tv_accvgpr_write_b32 a35, 0\n\tv_accvgpr_write_b32 a36, 0\n\tv_accvgpr_write_b32 a37, 0\n\tv_accvgpr_write_b32 a38, 0\n\tv_accvgpr_write_b32 a39, 0\n\tv_accvgpr_write_b32 a40, 0\n\tv_accvgpr_write_b32 a41, 0\n\tv_accvgpr_write_b32 a42, 0\n\tv_accvgpr_write_b32 a43, 0\n\tv_accvgpr_write_b32 a44, 0\n\tv_accvgpr_write_b32 a45, 0\n\tv_accvgpr_write_b32 a46, 0\n\tv_accvgpr_write_b32 a47, 0\n\tv_accvgpr_write_b32 a48, 0\n\tv_accvgpr_write_b32 a49, 0\n\tv_accvgpr_write_b32 a50, 0\n\tv_accvgpr_write_b32 a51, 0\n\tv_accvgpr_write_b32 a52, 0\n\tv_accvgpr_write_b32 a53, 0\n\tv_accvgpr_write_b32 a54, 0\n\tv_accvgpr_write_b32 a55, 0\n\tv_accvgpr_write_b32 a56, 0\n\tv_accvgpr_write_b32 a57, 0\n\tv_accvgpr_write_b32 a58, 0\n\tv_accvgpr_write_b32 a59, 0\n\tv_accvgpr_write_b32 a60, 0\n\tv_accvgpr_write_b32 a61, 0\n\tv_accvgpr_write_b32 a62, 0\n\tv_accvgpr_write_b32 a63, 0\n\tv_accvgpr_write_b32 a64, 0\n\tv_accvgpr_write_b32 a65, 0\n\tv_accvgpr_write_b32 a66, 0\n\tv_accvgpr_write_b32 a67, 0\n\tv_accvgpr_write_b32 a68, 0\n\tv_accvgpr_write_b32 a69, 0\n\tv_accvgpr_write_b32 a70, 0\n\tv_accvgpr_write_b32 a71, 0\n\tv_accvgpr_write_b32 a72, 0\n\tv_accvgpr_write_b32 a73, 0\n\tv_accvgpr_write_b32 a74, 0\n\tv_accvgpr_write_b32 a75, 0\n\tv_accvgpr_write_b32 a76, 0\n\tv_accvgpr_write_b32 a77, 0\n\tv_accvgpr_write_b32 a78, 0\n\tv_accvgpr_write_b32 a79, 0\n\tv_accvgpr_write_b32 a80, 0\n\tv_accvgpr_write_b32 a81, 0\n\tv_accvgpr_write_b32 a82, 0\n\tv_accvgpr_write_b32 a83, 0\n\tv_accvgpr_write_b32 a84, 0\n\tv_accvgpr_write_b32 a85, 0\n\tv_accvgpr_write_b32 a86, 0\n\tv_accvgpr_write_b32 a87, 0\n\tv_accvgpr_write_b32 a88, 0\n\tv_accvgpr_write_b32 a89, 0\n\tv_accvgpr_write_b32 a90, 0\n\tv_accvgpr_write_b32 a91, 0\n\tv_accvgpr_write_b32 a92, 0\n\tv_accvgpr_write_b32 a93, 0\n\tv_accvgpr_write_b32 a94, 0\n\tv_accvgpr_write_b32 a95, 0\n\t" ::: W_CLOB_ALL);      // O^T = 0; the clobber list is what reserves a[0:239] for the asm text
    int qi[2];
    bool q_ok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        qi[j] = q0 + 32 * j + r;
        q_ok[j] = qi[j] < Lq;
        qi[j] = q_ok[j] ? qi[j] : Lq - 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (pend_obase) store_rows(qo_region(par ^ 1), pend_obase, pend_q0);
#ifdef W_STAMP
    uint64_t t_landed;
    W_SB; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_landed) :: "memory"); W_SB;
#endif
    f32x16 sa[2][2], sb[2][2];
    bf16x8 pa[2][4], pb[2][4];
    w_for<0, 12>([&](auto I) { k_read(I, 0u); });          // (asm, into ACC registers: in flight under the Q arithmetic below)
    {   // Q^T fragments: lane (r, h) holds Q[q0 + 32 j + r][16 ks + 8 h .. + 7], read like a K fragment (query block j = key block kb)
        u32x4 uq[12];
#pragma unroll
        for (int I = 0; I < 12; ++I) {
            const int j = I / 6, ks = I % 6;
            const uint32_t a = (ks < 4 ? ka0 + 32 * ks : ks == 4 ? ka4 : ka5) - smem_a + qo_a + 32 * W_ROWB * j;
            uq[I] = *(lds_u4p)(uintptr_t)a;
        }
        // Q is scaled by scale * log2(e) once, here: the scores leave the MFMA chain in log2 units and, with the accumulators started at
        // -reference point, as the exponent itself -- no multiply-add per score in the loop (64 per lane and tile)
        auto scl = [&](uint32_t u) { return pack_bf16x2(lo16_to_f32(u) * scale_log2e, hi16_to_f32(u) * scale_log2e); };
        w_for<0, 12>([&](auto I) { w_qput<WA_Q + 4 * I>(make_uint4(scl(uq[I][0]), scl(uq[I][1]), scl(uq[I][2]), scl(uq[I][3]))); });
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 1" ::: "memory");
    w_for<0, 24>([&](auto I) { qk(I, sa, std::true_type{}); });      // S(0) from zero: its reference point is not known yet
    W_SB;
    if (nkt > 1) w_for<0, 12>([&](auto I) { k_read(I, (uint32_t)W_TILE); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // every wave has read K buffers 0 and 1
    if (!is_v && nkt > 2) dma(src_bh, 2);
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(sa[0][0]), "+v"(sa[0][1]), "+v"(sa[1][0]), "+v"(sa[1][1]));      // S(0) has left the matrix pipe
    {
        if (ragged) {                           // tile 0 is the ragged tail: keys >= Lk get -inf
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int key = (nkt - 1) * W_KT + 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * h;
                        sa[j][kb][i] = key < Lk_it ? sa[j][kb][i] : -INFINITY;
                    }
        }
        float mx[2];
        tile_max(sa, mx);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            m_run[j] = mx[j];                 // (log2 units: Q carries scale * log2 e)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) sa[j][kb][i] -= mx[j];
#pragma unroll
            for (int i = 0; i < 16; ++i) nref[j][i] = -mx[j];
        }
        // materialised here: nothing pads a compiler register copy in front of the asm MFMA that takes the tuple as C
        asm volatile("s_nop 4" : "+v"(nref[0]), "+v"(nref[1]));
        wf32x2 ps = {0.f, 0.f};
        ex_range(sa, pa, IC<0>{}, IC<0>{}, IC<16>{}, ps);      // query block 1 of tile 0 is step 0's phase-1 work
        l_run[0] = ps[0] + ps[1];
    }
    W_SB;

    // ---- one key tile: P(t) complete in pc; S(t+1) is produced in phase 1 and turned into P(t+1) (pn) in phase 2 -------------
#ifdef W_STAMP
    uint64_t tacc[4] = {0, 0, 0, 0}, tprev;
#define W_T0() { W_SB; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev) :: "memory"); W_SB; }
#define W_T(N) { uint64_t tn_; W_SB; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tn_) :: "memory"); W_SB; tacc[N] += tn_ - tprev; tprev = tn_; }
#else
#define W_T0()
#define W_T(N)
#endif
    // One key tile t.  On entry: P(t) of query block 0 complete in pc[0], S(t) of query block 1 still in so[1] (reference point in
    // nref).  phase 1: S(t+1) = K(t+1) Q^T -> sn   beside   V^T(t) fragment reads, exponentials of so[1] -> pc[1]
    //                 phase 2: O^T += V^T(t) P(t)^T   beside   K(t+2) fragment reads, row maxima of sn, exponentials of sn[0] -> pn[0]
    // NEXT: tile t+1 exists; KRD: tile t+2 exists.  Compile-time: the slots carry no branches.
    auto step = [&](bf16x8 (&pc)[2][4], bf16x8 (&pn)[2][4], f32x16 (&so)[2][2], f32x16 (&sn)[2][2], int t, auto next_tag, auto krd_tag) {
        constexpr bool NEXT = decltype(next_tag)::value, KRD = decltype(krd_tag)::value;
        using J0 = std::integral_constant<int, 0>;
        using J1 = std::integral_constant<int, 1>;
        W_T0()
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        W_T(0)
        // this wave's six LDS-DMA pieces of V(t+1) / K(t+3) go out one per slot in the second half of phase 1 (slots without fragment
        // reads).  Past the last tile the index is clamped: the piece lands in a buffer nobody reads any more.
        int dt = is_v ? t + 1 : t + 3;
        const uint32_t d_dst = __builtin_amdgcn_readfirstlane(ring_a + (dt & 1) * W_TILE);
        dt = dt < nkt ? dt : nkt - 1;
        const char* d_base = src_bh + (int64_t)tile_key0(dt) * W_ROWB;      // (steps only ever fetch full tiles: the ragged one is tile 0)
        auto dma_piece = [&](auto P) {
            constexpr int pc_ = P;
            dma1(d_base + (pc_ >= 3 ? 16 * W_ROWB : 0), g_off[pc_ % 3], d_dst + 1024 * pc_);
        };
        const uint32_t vb_off = (t & 1) * W_TILE, kb_off = (t & 1) * W_TILE;      // V(t); K(t+2) shares t's parity
        wf32x2 psA[2] = {{0.f, 0.f}, {0.f, 0.f}};
        if constexpr (NEXT) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // K(t+1) fragments (requested a phase ago)
            w_for<0, 24>([&](auto I) {
                if constexpr (!(W_ABL & 32)) qk(I, sn, std::false_type{});
                if constexpr (I < 12) { if constexpr (!(W_ABL & 8)) v_read(I, vb_off); }
                else if constexpr ((I & 1) == 0 && !(W_ABL & 16)) dma_piece(IC<(I - 12) / 2>{});
                // GENERATED PHASE1 BEGIN (tools/gen_w64_slots.py)
                if constexpr (I == 0) { ex_b0(so, J1{}, IC<0>{}, IC<0>{}); ex_b1(so, J1{}, IC<0>{}, IC<0>{}); }
                if constexpr (I == 1) { ex_b0(so, J1{}, IC<1>{}, IC<1>{}); ex_b1(so, J1{}, IC<1>{}, IC<1>{}); }
                if constexpr (I == 2) { ex_c(pc, J1{}, IC<0>{}, IC<0>{}, psA[1]); ex_b0(so, J1{}, IC<2>{}, IC<0>{}); ex_b1(so, J1{}, IC<2>{}, IC<0>{}); }
                if constexpr (I == 3) { ex_c(pc, J1{}, IC<1>{}, IC<1>{}, psA[1]); }
                if constexpr (I == 4) { ex_b0(so, J1{}, IC<3>{}, IC<1>{}); ex_b1(so, J1{}, IC<3>{}, IC<1>{}); }
                if constexpr (I == 5) { ex_c(pc, J1{}, IC<2>{}, IC<0>{}, psA[1]); ex_b0(so, J1{}, IC<4>{}, IC<0>{}); ex_b1(so, J1{}, IC<4>{}, IC<0>{}); }
                if constexpr (I == 6) { ex_c(pc, J1{}, IC<3>{}, IC<1>{}, psA[1]); }
                if constexpr (I == 7) { ex_b0(so, J1{}, IC<5>{}, IC<1>{}); ex_b1(so, J1{}, IC<5>{}, IC<1>{}); }
                if constexpr (I == 8) { ex_c(pc, J1{}, IC<4>{}, IC<0>{}, psA[1]); ex_b0(so, J1{}, IC<6>{}, IC<0>{}); ex_b1(so, J1{}, IC<6>{}, IC<0>{}); }
                if constexpr (I == 9) { ex_c(pc, J1{}, IC<5>{}, IC<1>{}, psA[1]); }
                if constexpr (I == 10) { ex_b0(so, J1{}, IC<7>{}, IC<1>{}); ex_b1(so, J1{}, IC<7>{}, IC<1>{}); }
                if constexpr (I == 11) { ex_c(pc, J1{}, IC<6>{}, IC<0>{}, psA[1]); ex_b0(so, J1{}, IC<8>{}, IC<0>{}); ex_b1(so, J1{}, IC<8>{}, IC<0>{}); }
                if constexpr (I == 12) { ex_c(pc, J1{}, IC<7>{}, IC<1>{}, psA[1]); }
                if constexpr (I == 13) { ex_b0(so, J1{}, IC<9>{}, IC<1>{}); ex_b1(so, J1{}, IC<9>{}, IC<1>{}); }
                if constexpr (I == 14) { ex_c(pc, J1{}, IC<8>{}, IC<0>{}, psA[1]); ex_b0(so, J1{}, IC<10>{}, IC<0>{}); ex_b1(so, J1{}, IC<10>{}, IC<0>{}); }
                if constexpr (I == 15) { ex_c(pc, J1{}, IC<9>{}, IC<1>{}, psA[1]); }
                if constexpr (I == 16) { ex_b0(so, J1{}, IC<11>{}, IC<1>{}); ex_b1(so, J1{}, IC<11>{}, IC<1>{}); }
                if constexpr (I == 17) { ex_c(pc, J1{}, IC<10>{}, IC<0>{}, psA[1]); ex_b0(so, J1{}, IC<12>{}, IC<0>{}); ex_b1(so, J1{}, IC<12>{}, IC<0>{}); }
                if constexpr (I == 18) { ex_c(pc, J1{}, IC<11>{}, IC<1>{}, psA[1]); }
                if constexpr (I == 19) { ex_b0(so, J1{}, IC<13>{}, IC<1>{}); ex_b1(so, J1{}, IC<13>{}, IC<1>{}); }
                if constexpr (I == 20) { ex_c(pc, J1{}, IC<12>{}, IC<0>{}, psA[1]); ex_b0(so, J1{}, IC<14>{}, IC<0>{}); ex_b1(so, J1{}, IC<14>{}, IC<0>{}); }
                if constexpr (I == 21) { ex_c(pc, J1{}, IC<13>{}, IC<1>{}, psA[1]); }
                if constexpr (I == 22) { ex_b0(so, J1{}, IC<15>{}, IC<1>{}); ex_b1(so, J1{}, IC<15>{}, IC<1>{}); }
                if constexpr (I == 23) { ex_c(pc, J1{}, IC<14>{}, IC<0>{}, psA[1]); ex_c(pc, J1{}, IC<15>{}, IC<1>{}, psA[1]); }
                // GENERATED PHASE1 END
                W_SB;
            });
        } else {
            w_for<0, 12>([&](auto I) { v_read(I, vb_off); });
            ex_range(so, pc, J1{}, IC<0>{}, IC<16>{}, psA[1]);
        }
        (void)d_dst; (void)d_base;
        l_run[0] += psA[0][0] + psA[0][1];
        l_run[1] += psA[1][0] + psA[1][1];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        W_SB;
        W_T(1)
        float alpha[2] = {1.f, 1.f}, dref[2] = {0.f, 0.f};
        bool moved = false;
        wf32x2 psB[2] = {{0.f, 0.f}, {0.f, 0.f}};
        float mq[2][4];
        w_for<0, 24>([&](auto I) {
            if constexpr (!(W_ABL & 64)) pv(I, pc);
            if constexpr (KRD && I < 12 && !(W_ABL & 8)) k_read(I, kb_off);
            if constexpr (NEXT && I >= 2 && I < 6 && (W_ABL & 4)) {
                if constexpr (I == 2) { mq[0][0] = mq[0][1] = mq[0][2] = mq[0][3] = mq[1][0] = mq[1][1] = mq[1][2] = mq[1][3] = 0.f; }
            }
            if constexpr (NEXT && I >= 2 && I < 6 && !(W_ABL & 4)) {
                // row maxima, four independent chains per query block, one quarter of the registers per slot.  (Not in slots 0, 1:
                // the last QK MFMAs of phase 1 are asm, nothing pads their results; two PV MFMAs later they have landed.)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int i = 4 * (I - 2) + c;
                        mq[j][c] = I == 2 ? fmaxf(sn[j][0][i], sn[j][1][i]) : fmaxf(fmaxf(mq[j][c], sn[j][0][i]), sn[j][1][i]);
                    }
                asm volatile("" : "+v"(mq[0][0]), "+v"(mq[0][1]), "+v"(mq[0][2]), "+v"(mq[0][3]), "+v"(mq[1][0]), "+v"(mq[1][1]), "+v"(mq[1][2]), "+v"(mq[1][3]));
            }
            if constexpr (NEXT && I == 6) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float m = fmaxf(fmaxf(mq[j][0], mq[j][1]), fmaxf(mq[j][2], mq[j][3]));
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
                    const float mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
                    // the reference point of a row's exponentials follows its maximum only when that jumps by more than 2^W_LAG:
                    // with 64 rows per wave SOME row's maximum moves in nearly every tile, and rescaling O^T means a round trip
                    // through the ACC registers; below the threshold P simply exceeds 1 (<= 2^W_LAG, exact in fp32 sums)
                    // (sn is already relative to the row's reference point: mx is the excess over it, in log2 units)
                    const bool jump = mx > W_LAG;
                    dref[j] = jump ? mx : 0.f;
                    alpha[j] = __builtin_amdgcn_exp2f(-dref[j]);
                    moved = moved || __any(jump);
                    m_run[j] += dref[j];
                }
                asm volatile("" : "+v"(dref[0]), "+v"(dref[1]), "+v"(alpha[0]), "+v"(alpha[1]));
                if (moved) {           // rare: this tile's scores and the accumulators' start value follow the reference point
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                            for (int i = 0; i < 16; ++i) sn[j][kb][i] -= dref[j];
#pragma unroll
                        for (int i = 0; i < 16; ++i) nref[j][i] -= dref[j];
                    }
                    asm volatile("s_nop 4" : "+v"(nref[0]), "+v"(nref[1]));
                }
            }
            if constexpr (NEXT) {                    // slots 7 .. 23: the 16 pairs of query block 0 of tile t+1
                // GENERATED PHASE2 BEGIN (tools/gen_w64_slots.py)
                if constexpr (I == 7) { ex_b0(sn, J0{}, IC<0>{}, IC<0>{}); ex_b1(sn, J0{}, IC<0>{}, IC<0>{}); ex_b0(sn, J0{}, IC<1>{}, IC<1>{}); }
                if constexpr (I == 8) { ex_b1(sn, J0{}, IC<1>{}, IC<1>{}); ex_c(pn, J0{}, IC<0>{}, IC<0>{}, psB[0]); ex_b0(sn, J0{}, IC<2>{}, IC<0>{}); }
                if constexpr (I == 9) { ex_b1(sn, J0{}, IC<2>{}, IC<0>{}); ex_c(pn, J0{}, IC<1>{}, IC<1>{}, psB[0]); ex_b0(sn, J0{}, IC<3>{}, IC<1>{}); }
                if constexpr (I == 10) { ex_b1(sn, J0{}, IC<3>{}, IC<1>{}); ex_c(pn, J0{}, IC<2>{}, IC<0>{}, psB[0]); ex_b0(sn, J0{}, IC<4>{}, IC<0>{}); }
                if constexpr (I == 11) { ex_b1(sn, J0{}, IC<4>{}, IC<0>{}); ex_c(pn, J0{}, IC<3>{}, IC<1>{}, psB[0]); }
                if constexpr (I == 12) { ex_b0(sn, J0{}, IC<5>{}, IC<1>{}); ex_b1(sn, J0{}, IC<5>{}, IC<1>{}); ex_c(pn, J0{}, IC<4>{}, IC<0>{}, psB[0]); }
                if constexpr (I == 13) { ex_b0(sn, J0{}, IC<6>{}, IC<0>{}); ex_b1(sn, J0{}, IC<6>{}, IC<0>{}); ex_c(pn, J0{}, IC<5>{}, IC<1>{}, psB[0]); }
                if constexpr (I == 14) { ex_b0(sn, J0{}, IC<7>{}, IC<1>{}); ex_b1(sn, J0{}, IC<7>{}, IC<1>{}); ex_c(pn, J0{}, IC<6>{}, IC<0>{}, psB[0]); }
                if constexpr (I == 15) { ex_b0(sn, J0{}, IC<8>{}, IC<0>{}); ex_b1(sn, J0{}, IC<8>{}, IC<0>{}); ex_c(pn, J0{}, IC<7>{}, IC<1>{}, psB[0]); }
                if constexpr (I == 16) { ex_b0(sn, J0{}, IC<9>{}, IC<1>{}); ex_b1(sn, J0{}, IC<9>{}, IC<1>{}); ex_c(pn, J0{}, IC<8>{}, IC<0>{}, psB[0]); }
                if constexpr (I == 17) { ex_b0(sn, J0{}, IC<10>{}, IC<0>{}); ex_b1(sn, J0{}, IC<10>{}, IC<0>{}); }
                if constexpr (I == 18) { ex_c(pn, J0{}, IC<9>{}, IC<1>{}, psB[0]); ex_b0(sn, J0{}, IC<11>{}, IC<1>{}); ex_b1(sn, J0{}, IC<11>{}, IC<1>{}); }
                if constexpr (I == 19) { ex_c(pn, J0{}, IC<10>{}, IC<0>{}, psB[0]); ex_b0(sn, J0{}, IC<12>{}, IC<0>{}); ex_b1(sn, J0{}, IC<12>{}, IC<0>{}); }
                if constexpr (I == 20) { ex_c(pn, J0{}, IC<11>{}, IC<1>{}, psB[0]); ex_b0(sn, J0{}, IC<13>{}, IC<1>{}); ex_b1(sn, J0{}, IC<13>{}, IC<1>{}); }
                if constexpr (I == 21) { ex_c(pn, J0{}, IC<12>{}, IC<0>{}, psB[0]); ex_b0(sn, J0{}, IC<14>{}, IC<0>{}); ex_b1(sn, J0{}, IC<14>{}, IC<0>{}); }
                if constexpr (I == 22) { ex_c(pn, J0{}, IC<13>{}, IC<1>{}, psB[0]); ex_b0(sn, J0{}, IC<15>{}, IC<1>{}); ex_b1(sn, J0{}, IC<15>{}, IC<1>{}); }
                if constexpr (I == 23) { ex_c(pn, J0{}, IC<14>{}, IC<0>{}, psB[0]); ex_c(pn, J0{}, IC<15>{}, IC<1>{}, psB[0]); }
                // GENERATED PHASE2 END
            }
            W_SB;
        });
        if constexpr (NEXT) {
            l_run[0] = l_run[0] * alpha[0] + psB[0][0] + psB[0][1];
            l_run[1] *= alpha[1];
            W_T(2)
#ifdef W_NORESCALE
            if (false) {
#else
            if (moved) {           // some query's reference point moved: O^T *= alpha (the PV MFMAs above have left the pipe first)
#endif
                asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
                w_for<0, 48>([&](auto R) { w_oscale<R>(alpha[0]); });
                w_for<48, 96>([&](auto R) { w_oscale<R>(alpha[1]); });
            }
        }
        W_T(3)
        W_SB;
    };
#ifdef W_STAMP
    uint64_t t_loop0;
    W_SB; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_loop0) :: "memory"); W_SB;
#endif
    {
        using T_ = std::true_type;
        using F_ = std::false_type;
        // steady steps (tiles t+1 and t+2 both present) run in pairs; an odd one left over swaps the roles of the two tail steps
        const int n_steady = nkt >= 2 ? nkt - 2 : 0;
        int t = 0;
        for (; t + 2 <= n_steady; t += 2) {
            step(pa, pb, sa, sb, t, T_{}, T_{});
            step(pb, pa, sb, sa, t + 1, T_{}, T_{});
        }
        if (nkt < 2) {
            step(pa, pb, sa, sb, 0, F_{}, F_{});
        } else if (t < n_steady) {
            step(pa, pb, sa, sb, t, T_{}, T_{});
            step(pb, pa, sb, sa, t + 1, T_{}, F_{});
            step(pa, pb, sa, sb, t + 2, F_{}, F_{});
        } else {
            step(pa, pb, sa, sb, t, T_{}, F_{});
            step(pb, pa, sb, sa, t + 1, F_{}, F_{});
        }
    }

#ifdef W_STAMP
    uint64_t t_loop1;
    W_SB; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_loop1) :: "memory"); W_SB;
#endif
    // ---- the next item's first loads go out here: every wave has left the key loop (barrier: the K / V ring is free), and the
    // epilogue below runs under their latency ----------------------------------------------------------------------------------------
    __builtin_amdgcn_s_barrier();
    const int v_next = v_item + (int)gridDim.x;
    const bool has_next = v_next < n_items;
    int n_qtile = 0, n_bhh = 0, n_q0 = 0;
    const bf16_t* n_Qb = nullptr;
    const char* n_src = nullptr;
    if (has_next) {
        item_of(v_next, n_qtile, n_bhh);
        n_q0 = n_qtile * W_QB + wave * 64;
        n_Qb = Q + (int64_t)n_bhh * Lq * 96;
        n_src = reinterpret_cast<const char*>((is_v ? V : Kt) + (int64_t)n_bhh * Lk_all * 96);
        issue_loads(n_src, n_Qb, n_q0, qo_region(par ^ 1));
    }
    // ---- epilogue: normalise, + q residual (the raw rows are still in this wave's LDS region), the 16-bit tile back into the same
    // slots, then whole rows out: store [b][q][g*96 + d] ------------------------------------------
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const int rot = (r >> 2) & 3;
    lds_u2p slot[12];                 // chunk 4 db + i4 of tile row r (query block 0; block 1 is 32 rows further), half h
    w_for<0, 12>([&](auto I) {
        int pc = I + rot;
        pc = pc >= 12 ? pc - 12 : pc;
        slot[I] = (lds_u2p)(uintptr_t)(qo_a + r * W_ROWB + pc * 16 + 8 * h);
    });
    u32x2 qres[2][12];                // all 24 residual reads go out before the first slot is overwritten
    if (ADD_Q && !(W_ABL & 128)) {
        w_for<0, 24>([&](auto I) { qres[I / 12][I % 12] = *(lds_u2p)((__attribute__((address_space(3))) char*)slot[I % 12] + 32 * W_ROWB * (I / 12)); });
    }
    w_for<0, 2>([&](auto J) {
        constexpr int j = J;
        const float l_tot = l_run[j] + __shfl_xor(l_run[j], 32, 64);
        const float inv = 1.0f / l_tot;
        if (LSE && q_ok[j] && h == 0) LSE[(int64_t)bh * Lq + qi[j]] = m_run[j] + __builtin_amdgcn_logf(l_tot);  // log2 domain
        w_for<0, 12>([&](auto I) {
            constexpr int db = I / 4, i4 = I % 4;                 // d = 32 db + 8 i4 + 4 h .. + 3: half h of 16-byte chunk 4 db + i4 = I
            float4 v;
            w_oget4<16 * (3 * j + db) + 4 * i4>(v);
            v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
            if (ADD_Q && !(W_ABL & 128)) {
                const u32x2 u = qres[j][I];
                v.x += lo16_to_f32(u[0]); v.y += hi16_to_f32(u[0]); v.z += lo16_to_f32(u[1]); v.w += hi16_to_f32(u[1]);
            }
            u32x2 o;
            o[0] = pack_bf16x2(v.x, v.y);
            o[1] = pack_bf16x2(v.z, v.w);
            *(lds_u2p)((__attribute__((address_space(3))) char*)slot[I] + 32 * W_ROWB * j) = o;
        });
    });
    {
        const int b = bh / heads, g = bh - b * heads;
        char* const obase = reinterpret_cast<char*>(O) + (((int64_t)b * Lq + q0) * (heads * 96) + g * 96) * 2;
        if (!has_next) {
            store_rows(qo_a, obase, q0);
        } else {                  // rows go out at the top of the next item, behind its operands
            pend_obase = obase;
            pend_q0 = q0;
        }
    }
#ifdef W_STAMP
    uint64_t t_end;
    W_SB; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_end) :: "memory"); W_SB;
    for (int i = 0; i < 4; ++i) sacc[i] += (float)tacc[i] / nkt;
    sacc[4] += (float)(t_landed - t_entry);
    sacc[5] += (float)(t_loop0 - t_landed);
    sacc[6] += (float)(t_end - t_loop1);
    sacc[7] += (float)(t_end - t_entry);
    ++n_it;
    if (!has_next && LSE_stamp && lane == 0) {   // cycles per tile in (wait + barrier, phase 1, phase 2, rescale + rest), then per item:
        // (top .. operands landed, rest of the prologue, key loop end .. epilogue end, all); one row of 8 per wave
        float* dst = LSE_stamp + (int64_t)bh * Lq + qtile * W_QB + wave * 8;
        for (int i = 0; i < 8; ++i) dst[i] = sacc[i] / n_it;
    }
#endif
    if (!has_next) break;
    v_item = v_next; qtile = n_qtile; bh = n_bhh; q0 = n_q0; Qb = n_Qb; src_bh = n_src;
    par ^= 1;
  }
}

int attn_fwd_w64_prepare() {      // 144 KiB of dynamic LDS
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_w64_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, W_SMEM) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_w64_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, W_SMEM) != hipSuccess)
        return MVIT_ELAUNCH;
    return MVIT_OK;
}

// launcher used by mvit_attention_fwd (attention.hip) when this form is selected
int attn_fwd_w64_launch(const void* q, const void* k, const void* v, void* out, float* lse, int B, int heads, int Lq, int Lk,
                        float scale_log2e, int add_q, hipStream_t st) {
    const int n_bh = B * heads, n_items = ((Lq + W_QB - 1) / W_QB) * n_bh;
    static DevInts ncu_tab;
    int n_wg = dev_cu_count(ncu_tab) & ~7;              // one workgroup per CU (512 registers per lane, 144 KiB of LDS); a multiple of 8: see item_of
    static const char* env = getenv("MVIT_ATT_W64_WGS");       // A/B runs: 0 = one workgroup per item (no prefetch across items)
    if (env) n_wg = atoi(env) > 0 ? (atoi(env) & ~7) : n_items;
    if (n_wg <= 0 || n_wg > n_items) n_wg = n_items;
    dim3 grid(n_wg);
    if (add_q)
        hipLaunchKernelGGL((attn_fwd_w64_kernel<true>), grid, dim3(256), W_SMEM, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out,
                           lse, heads, Lq, Lk, scale_log2e, nullptr, n_bh);
    else
        hipLaunchKernelGGL((attn_fwd_w64_kernel<false>), grid, dim3(256), W_SMEM, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out,
                           lse, heads, Lq, Lk, scale_log2e, nullptr, n_bh);
    return MVIT_OK;
}
