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
// Q is multiplied by scale * log2(e) once (and rounded to the 16-bit type again) and every score accumulator STARTS at minus the row's
// reference point, so the scores leave the MFMA chain as the exponent itself: a pair of scores costs two v_exp, two adds and a pack.
// Every MFMA is an asm statement followed by its share of the softmax arithmetic and a scheduling barrier, so the instruction
// stream is the one written here.  Hazards the compiler cannot see (it pads nothing around asm): an S tile written by asm MFMAs is
// first read a whole phase later; a P fragment written by compiler VALU is read by an asm MFMA at least one slot later.
#include "common.h"

#define W_KT 64
#define W_ROWB 192
#define W_TILE (W_KT * W_ROWB)           // 12 KiB
#define W_QB 256
#define W_SMEM (4 * W_TILE)              // K0 | K1 | V0 | V1
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
                                                              float* __restrict__ /*unused*/, int /*unused*/) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int qtile, bh;
    xcd_group_map(qtile, bh);
    const int Lk = Lk_all;
    const int64_t key0 = 0;
    const int b = bh / heads, g = bh - b * heads;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int q0 = qtile * W_QB + wave * 64;
    const bf16_t* Qb = Q + (int64_t)bh * Lq * 96;
    const bf16_t* Kb = Kt + ((int64_t)bh * Lk_all + key0) * 96;
    const bf16_t* Vb = V + ((int64_t)bh * Lk_all + key0) * 96;

    asm volatile("v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_write_b32 a1, 0\n\tv_accvgpr_write_b32 a2, 0\n\tv_accvgpr_write_b32 a3, 0\n\tv_accvgpr_write_b32 a4, 0\n\tv_accvgpr_write_b32 a5, 0\n\tv_accvgpr_write_b32 a6, 0\n\tv_accvgpr_write_b32 a7, 0\n\tv_accvgpr_write_b32 a8, 0\n\tv_accvgpr_write_b32 a9, 0\n\tv_accvgpr_write_b32 a10, 0\n\tv_accvgpr_write_b32 a11, 0\n\tv_accvgpr_write_b32 a12, 0\n\tv_accvgpr_write_b32 a13, 0\n\tv_accvgpr_write_b32 a14, 0\n\tv_accvgpr_write_b32 a15, 0\n\tv_accvgpr_write_b32 a16, 0\n\tv_accvgpr_write_b32 a17, 0\n\tv_accvgpr_write_b32 a18, 0\n\tv_accvgpr_write_b32 a19, 0\n\tv_accvgpr_write_b32 a20, 0\n\tv_accvgpr_write_b32 a21, 0\n\tv_accvgpr_write_b32 a22, 0\n\tv_accvgpr_write_b32 a23, 0\n\tv_accvgpr_write_b32 a24, 0\n\tv_accvgpr_write_b32 a25, 0\n\tv_accvgpr_write_b32 a26, 0\n\tv_accvgpr_write_b32 a27, 0\n\tv_accvgpr_write_b32 a28, 0\n\tv_accvgpr_write_b32 a29, 0\n\tv_accvgpr_write_b32 a30, 0\n\tv_accvgpr_write_b32 a31, 0\n\tv_accvgpr_write_b32 a32, 0\n\tv_accvgpr_write_b32 a33, 0\n\tv_accvgpr_write_b32 a34, 0\n\tv_accvgpr_write_b32 a35, 0\n\tv_accvgpr_write_b32 a36, 0\n\tv_accvgpr_write_b32 a37, 0\n\tv_accvgpr_write_b32 a38, 0\n\tv_accvgpr_write_b32 a39, 0\n\tv_accvgpr_write_b32 a40, 0\n\tv_accvgpr_write_b32 a41, 0\n\tv_accvgpr_write_b32 a42, 0\n\tv_accvgpr_write_b32 a43, 0\n\tv_accvgpr_write_b32 a44, 0\n\tv_accvgpr_write_b32 a45, 0\n\tv_accvgpr_write_b32 a46, 0\n\tv_accvgpr_write_b32 a47, 0\n\tv_accvgpr_write_b32 a48, 0\n\tv_accvgpr_write_b32 a49, 0\n\tv_accvgpr_write_b32 a50, 0\n\tv_accvgpr_write_b32 a51, 0\n\tv_accvgpr_write_b32 a52, 0\n\tv_accvgpr_write_b32 a53, 0\n\tv_accvgpr_write_b32 a54, 0\n\tv_accvgpr_write_b32 a55, 0\n\tv_accvgpr_write_b32 a56, 0\n\tv_accvgpr_write_b32 a57, 0\n\tv_accvgpr_write_b32 a58, 0\n\tv_accvgpr_write_b32 a59, 0\n\tv_accvgpr_write_b32 a60, 0\n\tv_accvgpr_write_b32 a61, 0\n\tv_accvgpr_write_b32 a62, 0\n\tv_accvgpr_write_b32 a63, 0\n\tv_accvgpr_write_b32 a64, 0\n\tv_accvgpr_write_b32 a65, 0\n\tv_accvgpr_write_b32 a66, 0\n\tv_accvgpr_write_b32 a67, 0\n\tv_accvgpr_write_b32 a68, 0\n\tv_accvgpr_write_b32 a69, 0\n\tv_accvgpr_write_b32 a70, 0\n\tv_accvgpr_write_b32 a71, 0\n\tv_accvgpr_write_b32 a72, 0\n\tv_accvgpr_write_b32 a73, 0\n\tv_accvgpr_write_b32 a74, 0\n\tv_accvgpr_write_b32 a75, 0\n\tv_accvgpr_write_b32 a76, 0\n\tv_accvgpr_write_b32 a77, 0\n\tv_accvgpr_write_b32 a78, 0\n\tv_accvgpr_write_b32 a79, 0\n\tv_accvgpr_write_b32 a80, 0\n\tv_accvgpr_write_b32 a81, 0\n\tv_accvgpr_write_b32 a82, 0\n\tv_accvgpr_write_b32 a83, 0\n\tv_accvgpr_write_b32 a84, 0\n\tv_accvgpr_write_b32 a85, 0\n\tv_accvgpr_write_b32 a86, 0\n\tv_accvgpr_write_b32 a87, 0\n\tv_accvgpr_write_b32 a88, 0\n\tv_accvgpr_write_b32 a89, 0\n\tv_accvgpr_write_b32 a90, 0\n\tv_accvgpr_write_b32 a91, 0\n\tv_accvgpr_write_b32 a92, 0\n\tv_accvgpr_write_b32 a93, 0\n\tv_accvgpr_write_b32 a94, 0\n\tv_accvgpr_write_b32 a95, 0\n\t" ::: W_CLOB_ALL);      // O^T = 0; the clobber list is what reserves a[0:239] for the asm text
    int qi[2];
    bool q_ok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        qi[j] = q0 + 32 * j + r;
        q_ok[j] = qi[j] < Lq;
        qi[j] = q_ok[j] ? qi[j] : Lq - 1;
    }
    // Q^T fragments: lane (r, h) holds Q[q0 + 32 j + r][16 ks + 8 h .. + 7]
    {   // all twelve fragment loads in flight together, then the ACC writes
        uint4 uq[12];
#pragma unroll
        for (int I = 0; I < 12; ++I) uq[I] = *reinterpret_cast<const uint4*>(Qb + (int64_t)qi[I / 6] * 96 + 16 * (I % 6) + 8 * h);
        // Q is scaled by scale * log2(e) once, here: the scores leave the MFMA chain in log2 units and, with the accumulators started at
        // -reference point, as the exponent itself -- no multiply-add per score in the loop (64 per lane and tile)
        auto scl = [&](uint32_t u) { return pack_bf16x2(lo16_to_f32(u) * scale_log2e, hi16_to_f32(u) * scale_log2e); };
#pragma unroll
        for (int I = 0; I < 12; ++I) uq[I] = make_uint4(scl(uq[I].x), scl(uq[I].y), scl(uq[I].z), scl(uq[I].w));
        w_for<0, 12>([&](auto I) { w_qput<WA_Q + 4 * I>(uq[I]); });
    }

    // LDS-DMA: waves 0, 1 move K tiles, waves 2, 3 V tiles, six 1-KiB pieces each (layout and swizzle as in attention.hip)
    const bool is_v = wave >= 2;
    const char* src_bh = reinterpret_cast<const char*>(is_v ? Vb : Kb);
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
    auto dma = [&](int tile) {               // this wave's six pieces of K / V tile `tile` into buffer tile & 1
        const int k0 = tile_key0(tile);
        const char* t_base = src_bh + (int64_t)k0 * W_ROWB;
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
        e_p[k][0] = __builtin_amdgcn_exp2f(s[j][kb][8 * sh + 2 * jj]);
        asm volatile("" : "+v"(e_p[k][0]));
    };
    auto ex_b1 = [&](f32x16 (&s)[2][2], auto J, auto E, auto K) {
        constexpr int j = J, e = E, k = K, u = e / 4, jj = e % 4, kb = u / 2, sh = u % 2;
        e_p[k][1] = __builtin_amdgcn_exp2f(s[j][kb][8 * sh + 2 * jj + 1]);
        asm volatile("" : "+v"(e_p[k][1]));
    };
    auto ex_c = [&](bf16x8 (&pf)[2][4], auto J, auto E, auto K, wf32x2& ps) {             // row sum, pack; fourth word closes the fragment
        constexpr int j = J, e = E, k = K, u = e / 4, jj = e % 4;
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

    // ---- prologue: K(0), K(1), V(0) in; S(0), its softmax; K(1) fragments; K(2) on its way ---------------------------------
    dma(0);
    if (!is_v && nkt > 1) dma(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    f32x16 sa[2][2], sb[2][2];
    bf16x8 pa[2][4], pb[2][4];
    w_for<0, 12>([&](auto I) { k_read(I, 0u); });
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 1" ::: "memory");
    w_for<0, 24>([&](auto I) { qk(I, sa, std::true_type{}); });      // S(0) from zero: its reference point is not known yet
    W_SB;
    if (nkt > 1) w_for<0, 12>([&](auto I) { k_read(I, (uint32_t)W_TILE); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // every wave has read K buffers 0 and 1
    if (!is_v && nkt > 2) dma(2);
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
                        sa[j][kb][i] = key < Lk ? sa[j][kb][i] : -INFINITY;
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
    float dbg_moved = 0.f, dbg_mx = 0.f;
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
                qk(I, sn, std::false_type{});
                if constexpr (I < 12) v_read(I, vb_off);
                else if constexpr ((I & 1) == 0) dma_piece(IC<(I - 12) / 2>{});
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
            pv(I, pc);
            if constexpr (KRD && I < 12) k_read(I, kb_off);
            if constexpr (NEXT && I >= 2 && I < 6) {
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
#ifdef W_STAMP
            if (moved) dbg_moved += 1.f;
            dbg_mx = alpha[0];
#endif
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
    if (LSE && lane == 0) {       // diagnostic build: cycles per tile in (wait + barrier, phase 1, phase 2, rescale + rest), one row per wave
        for (int i = 0; i < 4; ++i) LSE[(int64_t)bh * Lq + qtile * W_QB + wave * 8 + i] = (float)tacc[i] / nkt;
        LSE[(int64_t)bh * Lq + qtile * W_QB + wave * 8 + 4] = dbg_moved;
        LSE[(int64_t)bh * Lq + qtile * W_QB + wave * 8 + 5] = dbg_mx;
        LSE[(int64_t)bh * Lq + qtile * W_QB + wave * 8 + 6] = m_run[0];
        LSE[(int64_t)bh * Lq + qtile * W_QB + wave * 8 + 7] = l_run[0];
        return;
    }
#endif
    // ---- epilogue: normalise, + q residual, store [b][q][g*96 + d] -------------------------------
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    w_for<0, 2>([&](auto J) {
        constexpr int j = J;
        const float l_tot = l_run[j] + __shfl_xor(l_run[j], 32, 64);
        const float inv = 1.0f / l_tot;
        if (LSE && q_ok[j] && h == 0) LSE[(int64_t)bh * Lq + qi[j]] = m_run[j] + __builtin_amdgcn_logf(l_tot);  // log2 domain
        const int C = heads * 96;
        bf16_t* orow = O + ((int64_t)b * Lq + qi[j]) * C + g * 96;
        const bf16_t* qrow = Qb + (int64_t)qi[j] * 96;
        w_for<0, 12>([&](auto I) {
            constexpr int db = I / 4, i4 = I % 4;
            float4 v;
            w_oget4<16 * (3 * j + db) + 4 * i4>(v);
            const int d = 32 * db + 8 * i4 + 4 * h;
            v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
            if (ADD_Q) {
                const float4 qq = load4(qrow + d);
                v.x += qq.x; v.y += qq.y; v.z += qq.z; v.w += qq.w;
            }
            if (q_ok[j]) store4(orow + d, v);
        });
    });
}

int attn_fwd_w64_prepare() { return MVIT_OK; }      // 48 KiB of dynamic LDS: no attribute needed

// launcher used by mvit_attention_fwd (attention.hip) when this form is selected
int attn_fwd_w64_launch(const void* q, const void* k, const void* v, void* out, float* lse, int B, int heads, int Lq, int Lk,
                        float scale_log2e, int add_q, hipStream_t st) {
    dim3 grid((Lq + W_QB - 1) / W_QB, B * heads);
    if (add_q)
        hipLaunchKernelGGL((attn_fwd_w64_kernel<true>), grid, dim3(256), W_SMEM, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out,
                           lse, heads, Lq, Lk, scale_log2e, nullptr, 0);
    else
        hipLaunchKernelGGL((attn_fwd_w64_kernel<false>), grid, dim3(256), W_SMEM, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out,
                           lse, heads, Lq, Lk, scale_log2e, nullptr, 0);
    return MVIT_OK;
}
