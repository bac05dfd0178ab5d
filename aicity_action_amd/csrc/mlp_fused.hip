// The block tail in ONE kernel (inference):  out = x + fc2(GELU_erf(fc1(LayerNorm(x))))
// (reference: slowfast/models/attention.py:436-445 -- x + drop_path(mlp(norm2(x))) -- with Mlp.forward of
// slowfast/models/common.py:26-34; drop-path is the identity in eval mode.)  The [M][4C] hidden never reaches HBM, norm2 is the
// kernel's prologue, the fp32 residual its epilogue.  Same skeleton as attention_w64.hip with "keys" = hidden units:
//     H^T = W1c . Xn^T        (32 hidden units of a chunk x the wave's tokens, K = C)
//     G^T = GELU(H^T + b1)    in registers: the C-layout of H^T IS the B-operand layout of the next product
//     Y^T += W2c . G^T        (C channels x the wave's tokens, K = 32)
// A wave owns 32 TB tokens for ALL C channels: Y^T (C/32 x TB tiles of 32x32, fp32) and the normalised token rows Xn^T (2C/32 x TB
// B-operand fragments, 16 bit) live in registers for the whole kernel -- (C, TB) = (384, 1), (192, 2): 192 + 96 registers, (96, 2):
// 96 + 48 -- so the waves of a workgroup share nothing but the read-only weight stream: no exchange of G between waves, no token tile
// in LDS, one s_barrier per chunk (ring hand-over).  A workgroup = 4 waves = 128 TB tokens, one per CU (launch bounds (256, 1)).
//
// Weights arrive PRE-PACKED (mvit_mlp_fused_pack, once per weight version): per chunk of 32 hidden units one contiguous block
//     [ W1 image: C/64 slabs of 32 rows x 128 B | W2 image: C rows x 64 B ]      (64 C + 64 C bytes)
// already in the LDS layout (XOR-swizzled 16-byte pieces, conflict-free for the ds_read_b128 row fragments), LayerNorm's gamma
// folded into W1 and beta into b1 (b1' = b1 + W1 beta): the kernel normalises with (x - mean) * rstd only, and every LDS-DMA
// piece is a linear 1-KiB copy (lane l <- base + 16 l).  Row m of a W1 chunk holds hidden unit 32c + swap23(m) (bits 2 and 3 of
// m exchanged): with that order the accumulator registers 8s..8s+7 of a lane are hidden units 16s + 8h + 0..7 -- exactly the
// B-operand fragment of k-step s of the second product, so GELU packs pairs of registers and nothing is permuted.
//
// LDS: W1 ring (3 chunks) | W2 ring (3 chunks) | b1' (fp32).  Iteration t (one chunk):
//     s_waitcnt vmcnt / s_barrier                 chunk data issued two iterations ago is visible; buffers read last iteration are free
//     phase 1:  H^T(t+1) = W1(t+1) Xn^T           beside GELU of the upper 16 hidden units of chunk t, LDS-DMA pieces
//     phase 2:  Y^T += W2(t) G^T(t)               beside GELU of the lower 16 hidden units of chunk t+1, LDS-DMA pieces
// Waves 0, 1 issue the W1 pieces of chunk t+3, waves 2, 3 the W2 pieces of chunk t+2 (C/32 pieces per wave and iteration).  The
// fragment reads run D slots ahead of their MFMA as one stream through both phases with counted lgkmcnt waits.
// Every MFMA is an asm statement (the compiler never sees the ACC registers); hazards it cannot see: an H tile written by asm
// MFMAs is first read by GELU two slots after the chain's last MFMA; a G fragment written by VALU is read by an asm MFMA a phase later.
#include "common.h"

#ifdef MVIT_HALF_IS_FP16
#define MF_MFMA "v_mfma_f32_32x32x16_f16 "
#define MF_NST 5
#else
#define MF_MFMA "v_mfma_f32_32x32x16_bf16 "
#define MF_NST 4
#endif
#define MF_D 5               // fragment reads in flight ahead of their MFMA
#ifndef MF_NTOP
#define MF_NTOP 4            // GELU units run in front of phase 1's first MFMA (under the latency of the first fragment reads behind the barrier)
#endif
#define MF_NFB (MF_D + 1)
#define MF_SB __builtin_amdgcn_sched_barrier(0)
#define MF_CLOB_ALL "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"

template <int I, int N, typename F>
__device__ __forceinline__ void mf_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        mf_for<I + 1, N>(f);
    }
}
template <int N> using MIC = std::integral_constant<int, N>;

// ---- asm MFMA / LDS / ACC helpers -------------------------------------------------------------------------------------------
// W >= 0: the statement opens with s_waitcnt lgkmcnt(W) -- the fragment read this MFMA consumes (one statement: nothing is padded in between)
template <int XR, int W> __device__ __forceinline__ void mf_g1a(f32x16& h, bf16x8& w) {        // h += w . X (X fragment in a[XR:XR+3])
    if constexpr (W >= 0) asm volatile("s_waitcnt lgkmcnt(%c4)\n\t" MF_MFMA "%0, %1, a[%c2:%c3], %0" : "+v"(h), "+v"(w) : "i"(XR), "i"(XR + 3), "i"(W));
    else asm volatile(MF_MFMA "%0, %1, a[%c2:%c3], %0" : "+v"(h) : "v"(w), "i"(XR), "i"(XR + 3));
}
template <int XR, int W> __device__ __forceinline__ void mf_g1a0(f32x16& h, bf16x8& w) {       // h = w . X
    if constexpr (W >= 0) asm volatile("s_waitcnt lgkmcnt(%c4)\n\t" MF_MFMA "%0, %1, a[%c2:%c3], 0" : "=&v"(h), "+v"(w) : "i"(XR), "i"(XR + 3), "i"(W));
    else asm volatile(MF_MFMA "%0, %1, a[%c2:%c3], 0" : "=&v"(h) : "v"(w), "i"(XR), "i"(XR + 3));
}
template <int W> __device__ __forceinline__ void mf_g1v(f32x16& h, bf16x8& w, const bf16x8& x) {          // h += w . x (x in arch VGPRs)
    if constexpr (W >= 0) asm volatile("s_waitcnt lgkmcnt(%c3)\n\t" MF_MFMA "%0, %1, %2, %0" : "+v"(h), "+v"(w) : "v"(x), "i"(W));
    else asm volatile(MF_MFMA "%0, %1, %2, %0" : "+v"(h) : "v"(w), "v"(x));
}
template <int YR, int W> __device__ __forceinline__ void mf_g2(bf16x8& w, const bf16x8& g) {   // a[YR:YR+15] += w . g
    if constexpr (W >= 0) asm volatile("s_waitcnt lgkmcnt(%c4)\n\t" MF_MFMA "a[%c2:%c3], %0, %1, a[%c2:%c3]" : "+v"(w) : "v"(g), "i"(YR), "i"(YR + 15), "i"(W));
    else asm volatile(MF_MFMA "a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(w), "v"(g), "i"(YR), "i"(YR + 15));
}
template <int YR, int XR, int W> __device__ __forceinline__ void mf_p1a(bf16x8& w) {        // a[YR:YR+15] += w . X (X fragment in a[XR:XR+3])
    if constexpr (W >= 0) asm volatile("s_waitcnt lgkmcnt(%c5)\n\t" MF_MFMA "a[%c1:%c2], %0, a[%c3:%c4], a[%c1:%c2]" : "+v"(w) : "i"(YR), "i"(YR + 15), "i"(XR), "i"(XR + 3), "i"(W));
    else asm volatile(MF_MFMA "a[%c1:%c2], %0, a[%c3:%c4], a[%c1:%c2]" ::"v"(w), "i"(YR), "i"(YR + 15), "i"(XR), "i"(XR + 3));
}
template <int YR, int W> __device__ __forceinline__ void mf_p1v(bf16x8& w, const bf16x8& x) {   // a[YR:YR+15] += w . x (x in arch VGPRs)
    if constexpr (W >= 0) asm volatile("s_waitcnt lgkmcnt(%c4)\n\t" MF_MFMA "a[%c2:%c3], %0, %1, a[%c2:%c3]" : "+v"(w) : "v"(x), "i"(YR), "i"(YR + 15), "i"(W));
    else asm volatile(MF_MFMA "a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(w), "v"(x), "i"(YR), "i"(YR + 15));
}
template <int OFF> __device__ __forceinline__ void mf_rd(bf16x8& f, uint32_t addr) {
#ifdef MF_ABL_NORD
    asm volatile("" : "+v"(f) : "v"(addr));
#else
    asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(f) : "v"(addr), "i"(OFF));
#endif
}
template <int OFF> __device__ __forceinline__ void mf_rd4(f32x4& f, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(f) : "v"(addr), "i"(OFF));
}
template <int N> __device__ __forceinline__ void mf_wait(bf16x8& f) {
    asm volatile("s_waitcnt lgkmcnt(%c1)" : "+v"(f) : "i"(N));
}
template <int R> __device__ __forceinline__ void mf_aput(const uint4& u) {
    asm volatile("v_accvgpr_write_b32 a%c4, %0\n\tv_accvgpr_write_b32 a%c5, %1\n\tv_accvgpr_write_b32 a%c6, %2\n\tv_accvgpr_write_b32 a%c7, %3"
                 ::"v"(u.x), "v"(u.y), "v"(u.z), "v"(u.w), "i"(R), "i"(R + 1), "i"(R + 2), "i"(R + 3));
}
template <int R> __device__ __forceinline__ void mf_azero4() {
    asm volatile("v_accvgpr_write_b32 a%c0, 0\n\tv_accvgpr_write_b32 a%c1, 0\n\tv_accvgpr_write_b32 a%c2, 0\n\tv_accvgpr_write_b32 a%c3, 0"
                 ::"i"(R), "i"(R + 1), "i"(R + 2), "i"(R + 3));
}
template <int R> __device__ __forceinline__ void mf_aget4(float4& v) {
    asm volatile("v_accvgpr_read_b32 %0, a%c4\n\tv_accvgpr_read_b32 %1, a%c5\n\tv_accvgpr_read_b32 %2, a%c6\n\tv_accvgpr_read_b32 %3, a%c7"
                 : "=v"(v.x), "=v"(v.y), "=v"(v.z), "=v"(v.w) : "i"(R), "i"(R + 1), "i"(R + 2), "i"(R + 3));
}

// ---- compile-time schedule of the GELU work ------------------------------------------------------------------------------------
// A half-chunk (16 hidden units) is 4 TB pairs of accumulator registers per lane; a pair goes through MF_NST stages of ~24 issue
// cycles.  Units are ordered as a skewed pipeline (stage k of pair tau - k at time tau: the units of one time step are independent)
// and dealt evenly to the slots [S0, NS) of their phase.
struct MfUnit { int pair, stage; };
template <int NP>
constexpr MfUnit mf_unit(int n) {
    int c = 0;
    for (int tau = 0; tau < NP + MF_NST - 1; ++tau)
        for (int k = MF_NST - 1; k >= 0; --k) {
            const int p = tau - k;
            if (p < 0 || p >= NP) continue;
            if (c == n) return MfUnit{p, k};
            ++c;
        }
    return MfUnit{-1, -1};
}
// NTOP: the first NTOP units run at "slot -1": in front of the phase's first MFMA, under the LDS latency of its first fragment reads
template <int NP, int S0, int NS, int NTOP>
constexpr int mf_unit_slot(int n) { return n < NTOP ? -1 : S0 + ((n - NTOP) * (NS - S0)) / (NP * MF_NST - NTOP); }

template <int CB, int TB>
struct MfCfg {
    static constexpr int C = 32 * CB, K1S = 2 * CB;        // channels; k-steps of the first product
    static constexpr int G1 = K1S * TB, G2 = 2 * CB * TB;   // MFMAs per phase
    static constexpr int L = K1S + 2 * CB;                  // fragment reads per iteration (each feeds TB MFMAs)
    static constexpr int WU = 64 * C;                       // bytes of one W1 (= one W2) chunk image
    static constexpr int CHB = 2 * WU;                      // packed bytes per chunk
    static constexpr int NXF = K1S * TB;                    // X fragments per wave
    static constexpr int YREG = CB * TB * 16;
    static constexpr int XACC = (NXF < (256 - YREG) / 4) ? NXF : (256 - YREG) / 4;     // X fragments kept in ACC registers
    static constexpr int XV = NXF - XACC;                   // ... and in arch VGPRs
    static constexpr int NPW = CB;                          // LDS-DMA pieces per wave and iteration
    static constexpr int NPH = 4 * TB;                      // register pairs per half chunk and lane
    static constexpr int B1BYTES = (16 * C + 1023) / 1024 * 1024;     // b1' (fp32 [4C]) padded to whole KiB
    static constexpr int BPBYTES = (4 * C + 1023) / 1024 * 1024;      // proj bias (fp32 [C]) padded likewise (PROJ form)
    // prologue / epilogue staging (wave-private LDS area inside the rings): a pass moves one token block's rows x one column part
    static constexpr int CP = (C == 384) ? 2 : 1;           // column parts per row (a full 128 x 384 fp32 tile does not fit the LDS)
    static constexpr int SEG = 4 * C / CP;                  // bytes of a row segment per pass: 768, 768, 384
    static constexpr int PR = SEG / 16;                     // 16-byte pieces per row segment: 48, 48, 24
    static constexpr int RS = SEG + 16;                     // padded row pitch: rows 4 banks apart -> the lane-per-row reads / writes are conflict-free
    static constexpr int KSP = K1S / CP;                    // k-steps per pass
    static constexpr int ST_PITCH = (32 * RS + 255) / 256 * 256;
    static constexpr int NI = PR / 2;                       // 1-KiB wave instructions per pass on the linear side
    static constexpr int RPI = 192 / PR;                    // rows three such instructions advance
    static_assert(192 % PR == 0 && NI % 3 == 0, "column pattern of the linear side repeats every three instructions");
    static constexpr int PB1 = 0, PB2 = K1S;                // stream positions behind which the two bias reads of a phase are issued
    // lgkmcnt to wait for before consuming position n = the reads younger than read(n) issued by then: the fragment reads n+1 .. n+D-1
    // (as far as the stream goes) and the bias pair of a phase when it was issued in between (none in the last iteration)
    static constexpr int wcount(int n, bool with_bias) {
#ifdef MF_ABL_NORD
        return -1;
#endif
        const int younger = (L - 1 - n < MF_D - 1) ? L - 1 - n : MF_D - 1;
        const int b1c = (with_bias && PB1 < n && n <= PB1 + MF_D) ? 2 : 0;
        const int b2c = (with_bias && PB2 < n && n <= PB2 + MF_D) ? 2 : 0;
        return younger + b1c + b2c;
    }
    static_assert(PB1 + MF_D + 1 <= K1S, "the phase-1 bias reads must be older than a later fragment read of their phase");
    static_assert(L % MF_NFB == 0, "fragment ring");
};

// PROJ: the attention output projection rides along (attention.py:281,434: y = r + proj(o), then the block tail on y): X is the
// residual r, Oa the attention output (16 bit); Y^T starts at r, the proj product accumulates straight onto those accumulators (same
// W1-format chunk images, streamed through the W1 ring in front of the MLP's), + proj bias, and LayerNorm is taken FROM the
// accumulators.  y never reaches HBM and the proj launch (HBM-bound: 193 MB for 15 us of MFMA at stage 3) disappears.
template <int CB, int TB, bool PROJ>
__global__ __launch_bounds__(256, 1) void mlp_fused_kernel(const float* __restrict__ X, const bf16_t* __restrict__ Oa, const char* __restrict__ Wpk,
                                                           const float* __restrict__ B1p, const float* __restrict__ B2, float* __restrict__ Out,
                                                           int64_t M, int nch, float eps) {
    using K = MfCfg<CB, TB>;
    constexpr int NPC = CB;                               // proj chunks (32 output channels each)
    const char* const Wmlp = Wpk + (PROJ ? NPC * K::WU : 0);
    constexpr int C = K::C, K1S = K::K1S, WU = K::WU;
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef MF_STAMP
    uint64_t t_start;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_start) :: "memory");
#endif
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int64_t tok0 = (int64_t)blockIdx.x * (128 * TB) + wave * (32 * TB);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)(smem);
    const uint32_t lds_w1 = lds0, lds_w2 = lds0 + 3 * WU, lds_b1 = lds0 + 6 * WU;

    // the clobber list reserves every ACC register for the asm text (Y^T is initialised in the prologue below)
    asm volatile("" ::: MF_CLOB_ALL);

    // ---- LDS-DMA: waves 0, 1 move W1 chunk images, waves 2, 3 W2 chunk images; a piece is a linear 1-KiB copy ---------------------
    const bool is_w2 = wave >= 2;
    const char* src_role = Wmlp + (is_w2 ? WU : 0) + 1024 * (K::NPW * (wave & 1));
    const uint32_t dst_role = (is_w2 ? lds_w2 : lds_w1) + 1024 * (K::NPW * (wave & 1));
    uint32_t lane16 = 16u * lane;           // (not const: a generic lambda must capture it for its asm operand)
    auto dma1 = [&](const char* base, uint32_t lds) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(lane16), "s"(base) : "memory");
    };
    // piece P of a run of pieces 1 KiB apart in memory AND in LDS: the instruction's immediate offset moves both addresses, so a window
    // of four pieces shares its base registers
    auto dma_run = [&](const char* base, uint32_t lds, auto P_) {
        constexpr int P = P_;
        const uint32_t l16 = lane16;          // (an asm operand alone does not make a generic lambda capture the variable)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%c3" ::"s"(lds + 4096u * (P / 4)), "v"(l16),
                     "s"(base + 4096 * (P / 4)), "i"(1024 * (P % 4)) : "memory");
    };
    auto dma_chunk = [&](int chunk) {            // all of this wave's pieces of its kind's image of `chunk` (prologue)
        const char* b = src_role + (int64_t)chunk * K::CHB;
        const uint32_t d = __builtin_amdgcn_readfirstlane(dst_role + (uint32_t)(chunk % 3) * WU);
#pragma unroll
        for (int i = 0; i < K::NPW; ++i) dma1(b + 1024 * i, d + 1024 * i);
    };
    // ---- fragment addresses ---------------------------------------------------------------------------------------------------
    // W1 image: slab q (k 64q .. 64q+63) = 32 rows x 128 B, 16-byte piece (2 (ks % 4) + h) of row r at position piece ^ ((r >> 1) & 7);
    //           C = 96 ends in a half slab (k 64 .. 95) of 32 rows x 64 B, piece (2 (ks % 4) + h) at position piece ^ ((r >> 2) & 3)
    // W2 image: row ch x 64 B, piece (2 s + h) at position piece ^ ((ch >> 2) & 3)
    uint32_t a1[4], a2[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) a1[j] = lds_w1 + r * 128 + (((2 * j + h) ^ ((r >> 1) & 7)) << 4);
#pragma unroll
    for (int s = 0; s < 2; ++s) a2[s] = lds_w2 + r * 64 + (((2 * s + h) ^ ((r >> 2) & 3)) << 4);
    constexpr int NFS = C / 64;                  // full slabs; the half slab's lane addresses are those of the W2 image's rows
    const uint32_t ab = lds_b1 + 32 * h;             // bias piece (chunk c, half s, quarter q): + 128 c + 64 s + 16 q

    bf16x8 wf[MF_NFB];
    // stream position n of an iteration: n < K1S -> W1 fragment of k-step n (chunk unit u1), else W2 fragment j = n - K1S = s CB + cb (unit u2)
    auto frag_read = [&](auto N_, uint32_t u1, uint32_t u2) {
        constexpr int n = N_;
        if constexpr (n < 4 * NFS) mf_rd<(n / 4) * 4096>(wf[n % MF_NFB], a1[n % 4] + u1);
        else if constexpr (n < K1S) mf_rd<NFS * 4096>(wf[n % MF_NFB], a2[n % 4] - 3 * WU + u1);
        else mf_rd<((n - K1S) % CB) * 2048>(wf[n % MF_NFB], a2[(n - K1S) / CB] + u2);
    };
    // ---- prologue: the wave's token rows, LayerNorm statistics (two-pass, fp32), Xn^T fragments ------------------------------------
    // lane (r, h) of token block tb holds Xn[tok0 + 32 tb + r][16 ks + 8 h .. + 7] for every k-step ks.  Read lane-per-row straight
    // from memory every load instruction touched 32 cache lines for 32 bytes each (the prologue cost 29 k cycles of a 160 k-cycle
    // tile); here a pass brings 32 row segments in by LDS-DMA (one instruction per row, the lanes of a segment contiguous) into a
    // wave-private area with rows 16 bytes apart from a bank period, and the lanes read their rows from there.
    bf16x8 xv[K::XV > 0 ? K::XV : 1];
    char* const st_area = smem + wave * K::ST_PITCH;
    const uint32_t st_lds = lds0 + (uint32_t)wave * K::ST_PITCH;
    {
        // Y^T starts at x (the residual rides on the accumulators; the epilogue adds b2 and stores).  The lane reads channels 16 ks + 8 h + e of
        // its token, the accumulator layout wants 32 cb + 8 g + 4 h + i: the lane keeps its e = 0..3 (h = 0) / e = 4..7 (h = 1) and swaps the other
        // four with the lane of the other half (one v_permlane32_swap per register).  Fragment by fragment: nothing large is ever live in
        // arch VGPRs here -- under register pressure the compiler parks values in ACC registers it believes free, i.e. in the ones this
        // kernel's asm owns (tests/test_asm_kernel_audit.py).
        const uint64_t seg_mask = (K::PR >= 64) ? ~0ull : ((1ull << K::PR) - 1ull);
        mf_for<0, TB * K::CP>([&](auto P_) {
            constexpr int tb = P_ / K::CP, cp = P_ % K::CP;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // the previous pass's reads have left the area
#pragma unroll
            for (int row = 0; row < 32; ++row) {
                int64_t t = tok0 + 32 * tb + row;
                t = t < M ? t : M - 1;
                const char* src = reinterpret_cast<const char*>(X + t * C) + cp * K::SEG;
                const uint32_t dst = __builtin_amdgcn_readfirstlane(st_lds + row * K::RS);
                const uint32_t l16 = lane16;
                asm volatile("s_mov_b32 m0, %0\n\ts_mov_b64 exec, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, -1"
                             ::"s"(dst), "v"(l16), "s"(src), "s"(seg_mask) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (the issuing wave's own reads need no barrier behind its vmcnt)
            const char* rp = st_area + r * K::RS + 32 * h;
            mf_for<0, K::KSP>([&](auto KL_) {
                constexpr int ks = cp * K::KSP + KL_, cb = ks / 2, gh = ks % 2;
                const float4 lo = *reinterpret_cast<const float4*>(rp + 64 * KL_), hi = *reinterpret_cast<const float4*>(rp + 64 * KL_ + 16);
                const float lo_[4] = {lo.x, lo.y, lo.z, lo.w}, hi_[4] = {hi.x, hi.y, hi.z, hi.w};
                uint32_t ge[4], go[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo_[i]), __float_as_uint(hi_[i]), false, false);
                    ge[i] = sw[0];          // g = 2 gh:     channels 16 ks + 4 h + i
                    go[i] = sw[1];          // g = 2 gh + 1: channels 16 ks + 8 + 4 h + i
                }
                mf_aput<16 * (cb * TB + tb) + 4 * (2 * gh)>(make_uint4(ge[0], ge[1], ge[2], ge[3]));
                mf_aput<16 * (cb * TB + tb) + 4 * (2 * gh + 1)>(make_uint4(go[0], go[1], go[2], go[3]));
            });
        });
        if constexpr (PROJ) {        // the attention output rows (16 bit): one more pass per token block, straight into the fragment registers
            constexpr int RSO = 2 * C + 16, PRO = C / 8;
            const uint64_t o_mask = (1ull << PRO) - 1ull;
            mf_for<0, TB>([&](auto T_) {
                constexpr int tb = T_;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int row = 0; row < 32; ++row) {
                    int64_t t = tok0 + 32 * tb + row;
                    t = t < M ? t : M - 1;
                    const char* src = reinterpret_cast<const char*>(Oa + t * C);
                    const uint32_t dst = __builtin_amdgcn_readfirstlane(st_lds + row * RSO);
                    const uint32_t l16 = lane16;
                    asm volatile("s_mov_b32 m0, %0\n\ts_mov_b64 exec, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, -1"
                                 ::"s"(dst), "v"(l16), "s"(src), "s"(o_mask) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const char* rp = st_area + r * RSO + 16 * h;
                mf_for<0, K1S>([&](auto KS_) {
                    constexpr int ks = KS_, f = ks * TB + tb;
                    const uint4 u = *reinterpret_cast<const uint4*>(rp + 32 * ks);
                    if constexpr (f < K::XACC) mf_aput<K::YREG + 4 * f>(u);
                    else {
                        xv[f - K::XACC] = *reinterpret_cast<const bf16x8*>(&u);
                        asm volatile("" : "+v"(xv[f - K::XACC]));
                    }
                });
            });
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // every wave has read its rows: the area becomes the weight rings
        if constexpr (PROJ) {
            if (!is_w2) {                             // the W1-format stream S = [proj chunks | MLP chunks]: S[0], S[1]
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const char* b = Wpk + 1024 * (K::NPW * (wave & 1)) + c * WU;
                    const uint32_t d = __builtin_amdgcn_readfirstlane(dst_role + (uint32_t)c * WU);
#pragma unroll
                    for (int i = 0; i < K::NPW; ++i) dma1(b + 1024 * i, d + 1024 * i);
                }
            } else dma_chunk(0);
        } else dma_chunk(0);
        if (is_w2) {                                  // b1' (and the proj bias behind it): fp32, padded to whole KiB by the packer -> LDS, waves 2 and 3
            constexpr int NBP = (K::B1BYTES + (PROJ ? K::BPBYTES : 0)) / 1024, NB1W = (NBP + 1) / 2;
#pragma unroll
            for (int i = 0; i < NB1W; ++i) {
                int pc = (wave & 1) * NB1W + i;
                pc = pc < NBP ? pc : NBP - 1;
                dma1(reinterpret_cast<const char*>(B1p) + 1024 * pc, __builtin_amdgcn_readfirstlane(lds_b1 + 1024u * pc));
            }
        }
        if (!PROJ || is_w2) {
            if (nch > 1) dma_chunk(1);
            if (!is_w2 && nch > 2) dma_chunk(2);
        }
    }
    if constexpr (PROJ) {
        // ---- y = r + proj(o) on the accumulators: chunk p = output channels 32p .. 32p+31 = Y^T tile row p (+ b_proj below) -------------
        mf_for<0, NPC>([&](auto P_) {
            constexpr int p = P_;
            asm volatile("s_waitcnt vmcnt(%c0)" ::"i"(K::NPW) : "memory");
            __builtin_amdgcn_s_barrier();
            if (!is_w2) {                             // S[p + 2]: a proj chunk, or the W1 image of MLP chunk p + 2 - NPC
                constexpr int i2 = p + 2;
                const char* b = (i2 < NPC) ? Wpk + 1024 * (K::NPW * (wave & 1)) + i2 * WU : src_role + (int64_t)(i2 - NPC) * K::CHB;
                const uint32_t d = __builtin_amdgcn_readfirstlane(dst_role + (uint32_t)(i2 % 3) * WU);
#pragma unroll
                for (int i = 0; i < K::NPW; ++i) dma1(b + 1024 * i, d + 1024 * i);
            }
            constexpr uint32_t u1 = (uint32_t)(p % 3) * WU;
            mf_for<0, (MF_D < K1S ? MF_D : K1S)>([&](auto N_) { frag_read(N_, u1, 0u); });
            mf_for<0, K::G1>([&](auto I_) {
                constexpr int I = I_, n = I / TB, tb = I % TB, f = n * TB + tb;
                constexpr int W = (tb == 0) ? ((K1S - 1 - n < MF_D - 1) ? K1S - 1 - n : MF_D - 1) : -1;
                if constexpr (f < K::XACC) mf_p1a<16 * (p * TB + tb), K::YREG + 4 * f, W>(wf[n % MF_NFB]);
                else mf_p1v<16 * (p * TB + tb), W>(wf[n % MF_NFB], xv[f - K::XACC]);
                if constexpr (tb == 0 && n + MF_D < K1S) frag_read(MIC<n + MF_D>{}, u1, 0u);
                MF_SB;
            });
        });
        __builtin_amdgcn_s_barrier();                 // the last proj chunk's buffer is free: it takes the W1 image of MLP chunk 2
        if (!is_w2) {
            const char* b = src_role + (int64_t)2 * K::CHB;
            const uint32_t d = __builtin_amdgcn_readfirstlane(dst_role + 2u * WU);
#pragma unroll
            for (int i = 0; i < K::NPW; ++i) dma1(b + 1024 * i, d + 1024 * i);
        }
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");        // the last MFMAs have left the matrix pipe
    }
    // ---- LayerNorm FROM the accumulators (Y^T = the rows to normalise), streaming: sum | centred squares | fragments.  The accumulator
    // registers are read three times (a v_accvgpr_read each) instead of holding the row in arch VGPRs.  PROJ: + proj bias first, written back.
    {
        float mean_t[TB], rstd_t[TB];
        mf_for<0, TB>([&](auto T_) {
            constexpr int tb = T_;
            float sm = 0.f;
            mf_for<0, CB * 4>([&](auto Q_) {
                constexpr int cb = Q_ / 4, g = Q_ % 4;
                float4 a;
                mf_aget4<16 * (cb * TB + tb) + 4 * g>(a);
                if constexpr (PROJ) {
                    const float4 bb = *reinterpret_cast<const float4*>(smem + 6 * WU + K::B1BYTES + (32 * cb + 8 * g + 4 * h) * 4);
                    a.x += bb.x; a.y += bb.y; a.z += bb.z; a.w += bb.w;
                    mf_aput<16 * (cb * TB + tb) + 4 * g>(make_uint4(__float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(a.z), __float_as_uint(a.w)));
                }
                sm += (a.x + a.y) + (a.z + a.w);
            });
            sm += __shfl_xor(sm, 32, 64);
            const float mean = sm * (1.0f / C);
            float qq = 0.f;
            if constexpr (PROJ) asm volatile("s_nop 1" ::: "memory");       // (v_accvgpr_write -> v_accvgpr_read of the same register)
            mf_for<0, CB * 4>([&](auto Q_) {
                constexpr int cb = Q_ / 4, g = Q_ % 4;
                float4 a;
                mf_aget4<16 * (cb * TB + tb) + 4 * g>(a);
                a.x -= mean; a.y -= mean; a.z -= mean; a.w -= mean;
                qq = fmaf(a.x, a.x, qq); qq = fmaf(a.y, a.y, qq); qq = fmaf(a.z, a.z, qq); qq = fmaf(a.w, a.w, qq);
            });
            qq += __shfl_xor(qq, 32, 64);
            mean_t[tb] = mean;
            rstd_t[tb] = 1.0f / sqrtf(qq * (1.0f / C) + eps);
        });
        // lane (r, h) holds channels 32 cb + 8 g + 4 h + i; fragment k-step 2 cb + gh wants 32 cb + 16 gh + 8 h + e: the lane keeps
        // g = 2 gh + h and receives the other half's registers of the same g (v_permlane32_swap of the g-even with the g-odd register)
        mf_for<0, K::NXF>([&](auto F) {
            constexpr int f = F, ks = f / TB, tb = f % TB, cb = ks / 2, gh = ks % 2;
            float4 e, o;
            mf_aget4<16 * (cb * TB + tb) + 4 * (2 * gh)>(e);
            mf_aget4<16 * (cb * TB + tb) + 4 * (2 * gh + 1)>(o);
            const float m_ = mean_t[tb], rs_ = rstd_t[tb];
            const float ev[4] = {(e.x - m_) * rs_, (e.y - m_) * rs_, (e.z - m_) * rs_, (e.w - m_) * rs_};
            const float ov[4] = {(o.x - m_) * rs_, (o.y - m_) * rs_, (o.z - m_) * rs_, (o.w - m_) * rs_};
            float lo[4], hi[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(ev[i]), __float_as_uint(ov[i]), false, false);
                lo[i] = __uint_as_float(sw[0]);
                hi[i] = __uint_as_float(sw[1]);
            }
            const uint4 u = make_uint4(pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3]));
            if constexpr (f < K::XACC) mf_aput<K::YREG + 4 * f>(u);
            else {
                xv[f - K::XACC] = *reinterpret_cast<const bf16x8*>(&u);
                asm volatile("" : "+v"(xv[f - K::XACC]));
            }
        });
    }

    // ---- GELU of one register pair, in stages ------------------------------------------------------------------------------------
    // pair p of a half chunk s: token block p / 4, registers 8 s + 2 (p % 4), + 1 of its H tile; result = word p % 4 of G[tb][s]
    float gx[MF_NST][2], gq[MF_NST][2], g2[MF_NST][2];
    uint32_t gw[TB][2][4];
    auto gelu_unit = [&](f32x16 (&H)[TB], bf16x8 (&G)[TB][2], const f32x4 (&bs)[2], auto S_, auto P_, auto KS_) {
        constexpr int s = S_, p = P_, k = KS_, tb = p / 4, j = p % 4, v0 = 8 * s + 2 * j, sl = p % MF_NST;
        float(&x)[2] = gx[sl];
        float(&q)[2] = gq[sl];
        float(&t)[2] = g2[sl];
#ifdef MVIT_HALF_IS_FP16
        // degree-13 fit (7 coefficients): x, x^2, 6 fma | exp2 x 2, 1 + e | rcp x 2, x * Phi, pack  -- 5 stages of ~24 issue cycles
        if constexpr (k == 0) {
            x[0] = H[tb][v0] + bs[(2 * j) / 4][(2 * j) % 4];
            x[1] = H[tb][v0 + 1] + bs[(2 * j + 1) / 4][(2 * j + 1) % 4];
            t[0] = x[0] * x[0]; t[1] = x[1] * x[1];
            q[0] = fmaf(k_gelu[6], t[0], k_gelu[5]); q[1] = fmaf(k_gelu[6], t[1], k_gelu[5]);
            asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(t[0]), "+v"(t[1]), "+v"(q[0]), "+v"(q[1]));
        } else if constexpr (k == 1) {
            q[0] = fmaf(q[0], t[0], k_gelu[4]); q[1] = fmaf(q[1], t[1], k_gelu[4]);
            q[0] = fmaf(q[0], t[0], k_gelu[3]); q[1] = fmaf(q[1], t[1], k_gelu[3]);
            q[0] = fmaf(q[0], t[0], k_gelu[2]); q[1] = fmaf(q[1], t[1], k_gelu[2]);
            asm volatile("" : "+v"(q[0]), "+v"(q[1]));
        } else if constexpr (k == 2) {
            q[0] = fmaf(q[0], t[0], k_gelu[1]); q[1] = fmaf(q[1], t[1], k_gelu[1]);
            q[0] = fmaf(q[0], t[0], k_gelu[0]); q[1] = fmaf(q[1], t[1], k_gelu[0]);
            q[0] *= x[0]; q[1] *= x[1];
            asm volatile("" : "+v"(q[0]), "+v"(q[1]));
        } else if constexpr (k == 3) {
            q[0] = __builtin_amdgcn_exp2f(q[0]); q[1] = __builtin_amdgcn_exp2f(q[1]);
            q[0] = 1.0f + q[0]; q[1] = 1.0f + q[1];
            asm volatile("" : "+v"(q[0]), "+v"(q[1]));
        } else {
            q[0] = __builtin_amdgcn_rcpf(q[0]);
#else
        // degree-7 fit (4 coefficients): 4 stages of ~24 issue cycles
        if constexpr (k == 0) {
            x[0] = H[tb][v0] + bs[(2 * j) / 4][(2 * j) % 4];
            x[1] = H[tb][v0 + 1] + bs[(2 * j + 1) / 4][(2 * j + 1) % 4];
            t[0] = x[0] * x[0]; t[1] = x[1] * x[1];
            q[0] = fmaf(k_gelu[3], t[0], k_gelu[2]); q[1] = fmaf(k_gelu[3], t[1], k_gelu[2]);
            q[0] = fmaf(q[0], t[0], k_gelu[1]); q[1] = fmaf(q[1], t[1], k_gelu[1]);
            asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(t[0]), "+v"(t[1]), "+v"(q[0]), "+v"(q[1]));
        } else if constexpr (k == 1) {
            q[0] = fmaf(q[0], t[0], k_gelu[0]); q[1] = fmaf(q[1], t[1], k_gelu[0]);
            q[0] *= x[0]; q[1] *= x[1];
            q[0] = __builtin_amdgcn_exp2f(q[0]);
            asm volatile("" : "+v"(q[0]), "+v"(q[1]));
        } else if constexpr (k == 2) {
            q[1] = __builtin_amdgcn_exp2f(q[1]);
            q[0] = 1.0f + q[0]; q[1] = 1.0f + q[1];
            q[0] = __builtin_amdgcn_rcpf(q[0]);
            asm volatile("" : "+v"(q[0]), "+v"(q[1]));
        } else {
#endif
            q[1] = __builtin_amdgcn_rcpf(q[1]);
            gw[tb][s][j] = pack_bf16x2(x[0] * q[0], x[1] * q[1]);
            asm volatile("" : "+v"(gw[tb][s][j]));
            if constexpr (j == 3) {
                const uint4 u = make_uint4(gw[tb][s][0], gw[tb][s][1], gw[tb][s][2], gw[tb][s][3]);
                G[tb][s] = *reinterpret_cast<const bf16x8*>(&u);
                asm volatile("" : "+v"(G[tb][s]));       // the fragment exists from here on: no register copy lands in front of the asm MFMA reading it
            }
        }
    };
    // the units of half s that fall into slot I of a phase of NS slots starting at S0
    auto gelu_slot = [&](f32x16 (&H)[TB], bf16x8 (&G)[TB][2], const f32x4 (&bs)[2], auto S_, auto I_, auto S0_, auto NS_, auto NTOP_) {
        constexpr int I = I_, S0 = S0_, NS = NS_, NTOP = NTOP_;
#ifdef MF_ABL_NOGELU
        return;
#endif
        mf_for<0, K::NPH * MF_NST>([&](auto N_) {
            constexpr int n = N_;
            if constexpr (mf_unit_slot<K::NPH, S0, NS, NTOP>(n) == I) {
                constexpr MfUnit u = mf_unit<K::NPH>(n);
                gelu_unit(H, G, bs, S_, MIC<u.pair>{}, MIC<u.stage>{});
            }
        });
    };
    auto gelu_all = [&](f32x16 (&H)[TB], bf16x8 (&G)[TB][2], const f32x4 (&bs)[2], auto S_) {     // straight-line (prologue, last chunk)
        mf_for<0, K::NPH * MF_NST>([&](auto N_) {
            constexpr MfUnit u = mf_unit<K::NPH>(N_);
            gelu_unit(H, G, bs, S_, MIC<u.pair>{}, MIC<u.stage>{});
        });
    };

    // first-product MFMA of stream position n (k-step n), token block tb; W_ = lgkmcnt to wait for first (-1: none)
    auto g1 = [&](f32x16 (&H)[TB], auto N_, auto TB_, auto W_) {
        constexpr int n = N_, tb = TB_, f = n * TB + tb, W = W_;
        if constexpr (n == 0) mf_g1a0<K::YREG + 4 * f, W>(H[tb], wf[n % MF_NFB]);
        else if constexpr (f < K::XACC) mf_g1a<K::YREG + 4 * f, W>(H[tb], wf[n % MF_NFB]);
        else mf_g1v<W>(H[tb], wf[n % MF_NFB], xv[f - K::XACC]);
    };
    // second-product MFMA of stream position n = K1S + s CB + cb, token block tb
    auto g2m = [&](bf16x8 (&G)[TB][2], auto N_, auto TB_, auto W_) {
        constexpr int n = N_, tb = TB_, j = n - K1S, s = j / CB, cb = j % CB, W = W_;
        mf_g2<16 * (cb * TB + tb), W>(wf[n % MF_NFB], G[tb][s]);
    };

    f32x16 Ha[TB], Hb[TB];
    bf16x8 Ga[TB][2], Gb[TB][2];
    f32x4 bs0[2], bs1[2];           // b1' of the lower / upper 16 hidden units of the chunk whose GELU comes next

    // ---- before the loop: H(0) = W1(0) Xn^T, the GELU of its lower half ------------------------------------------------------------
    // chunk 0 (and b1') has landed; W1(1), W1(2) / W2(1) stay in flight
    if (is_w2) asm volatile("s_waitcnt vmcnt(%c0)" ::"i"(K::NPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(%c0)" ::"i"(2 * K::NPW) : "memory");
    __builtin_amdgcn_s_barrier();
    {
        mf_rd4<0>(bs0[0], ab);
        mf_rd4<16>(bs0[1], ab);
        mf_rd4<64>(bs1[0], ab);
        mf_rd4<80>(bs1[1], ab);
        mf_for<0, K1S>([&](auto N_) { frag_read(N_, 0u, 0u); if constexpr ((N_ % MF_NFB) == MF_NFB - 1 || N_ == K1S - 1) {
            // the ring holds MF_NFB fragments: consume a batch before reading on
            constexpr int n1 = N_, n0 = n1 - (n1 % MF_NFB);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            mf_for<n0, n1 + 1>([&](auto Q_) { asm volatile("" : "+v"(wf[Q_ % MF_NFB])); mf_for<0, TB>([&](auto T_) { g1(Ha, Q_, T_, MIC<-1>{}); }); });
        } });
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) asm volatile("" : "+v"(Ha[tb]));
        asm volatile("" : "+v"(bs0[0]), "+v"(bs0[1]), "+v"(bs1[0]), "+v"(bs1[1]));
        gelu_all(Ha, Ga, bs0, MIC<0>{});
    }
    MF_SB;

    // ---- one chunk t.  On entry: H(t) in Hc with its lower half already in Gc[.][0]; bs1 = upper-half bias of chunk t ---------------
#ifdef MF_STAMP
    uint64_t tacc[3] = {0, 0, 0}, tprev, t_loop0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_loop0) :: "memory");
#define MF_T0() { MF_SB; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev) :: "memory"); MF_SB; }
#define MF_T(N) { uint64_t tn_; MF_SB; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tn_) :: "memory"); MF_SB; tacc[N] += tn_ - tprev; tprev = tn_; }
#else
#define MF_T0()
#define MF_T(N)
#endif
    auto step = [&](f32x16 (&Hc)[TB], f32x16 (&Hn)[TB], bf16x8 (&Gc)[TB][2], bf16x8 (&Gn)[TB][2], int t, auto next_tag) {
        constexpr bool NEXT = decltype(next_tag)::value;
        // data of chunks t+1 (W1) / t (W2) was issued two iterations ago; leave last iteration's pieces in flight
        MF_T0()
        asm volatile("s_waitcnt vmcnt(%c0)" ::"i"(K::NPW) : "memory");
#ifndef MF_ABL_NOBAR
        __builtin_amdgcn_s_barrier();
#endif
        MF_T(0)
        const uint32_t u1 = (uint32_t)((t + 1) % 3) * WU, u2 = (uint32_t)(t % 3) * WU;
        // this wave's pieces of this iteration: W1 image of chunk t+3 / W2 image of chunk t+2 (clamped past the end: lands in a free buffer)
        int dc = is_w2 ? t + 2 : t + 3;
        const uint32_t d_dst = __builtin_amdgcn_readfirstlane(dst_role + (uint32_t)(dc % 3) * WU);
        dc = dc < nch ? dc : nch - 1;
        const char* d_src = src_role + (int64_t)dc * K::CHB;
#ifdef MF_ABL_NODMA
        auto dma_piece = [&](auto P_) { (void)d_src; (void)d_dst; };
#else
        auto dma_piece = [&](auto P_) { dma_run(d_src, d_dst, P_); };
#endif
        constexpr int NSLOT = (NEXT ? K::G1 : 0) + K::G2;
        constexpr int DSTRIDE = NSLOT / K::NPW;              // one piece every DSTRIDE slots
        const uint32_t abt = ab + 128u * (uint32_t)(t + 1);
        if constexpr (NEXT) {
            mf_for<0, MF_D>([&](auto N_) { frag_read(N_, u1, u2); });
            gelu_slot(Hc, Gc, bs1, MIC<1>{}, MIC<-1>{}, MIC<0>{}, MIC<K::G1>{}, MIC<MF_NTOP>{});      // under the latency of the reads just issued
            MF_SB;
            mf_for<0, K::G1>([&](auto I_) {
                constexpr int I = I_, n = I / TB, tb = I % TB;
                g1(Hn, MIC<n>{}, MIC<tb>{}, MIC<(tb == 0 ? K::wcount(n, true) : -1)>{});
                if constexpr (tb == 0 && n + MF_D < K::L) frag_read(MIC<n + MF_D>{}, u1, u2);
                if constexpr (tb == 0 && n == K::PB1) {          // b1' of the lower half of chunk t+1 (used in phase 2)
                    mf_rd4<0>(bs0[0], abt);
                    mf_rd4<16>(bs0[1], abt);
                }
                if constexpr (I % DSTRIDE == DSTRIDE / 2 && I / DSTRIDE < K::NPW) dma_piece(MIC<I / DSTRIDE>{});
                gelu_slot(Hc, Gc, bs1, MIC<1>{}, I_, MIC<0>{}, MIC<K::G1>{}, MIC<MF_NTOP>{});
                MF_SB;
            });
        } else {
            mf_for<K1S, K1S + MF_D>([&](auto N_) { frag_read(N_, u1, u2); });
            gelu_all(Hc, Gc, bs1, MIC<1>{});
        }
        asm volatile("" : "+v"(bs0[0]), "+v"(bs0[1]));
        MF_T(1)
        mf_for<0, K::G2>([&](auto I_) {
            constexpr int I = I_, n = K1S + I / TB, tb = I % TB;
            g2m(Gc, MIC<n>{}, MIC<tb>{}, MIC<(tb == 0 ? K::wcount(n, NEXT) : -1)>{});      // (last chunk: the stream started at K1S, no bias reads)
            if constexpr (tb == 0 && n + MF_D < K::L) frag_read(MIC<n + MF_D>{}, u1, u2);
            if constexpr (NEXT && tb == 0 && n == K::PB2) {      // b1' of the upper half of chunk t+1 (used in the next phase 1)
                mf_rd4<64>(bs1[0], abt);
                mf_rd4<80>(bs1[1], abt);
            }
            constexpr int IS = (NEXT ? K::G1 : 0) + I;
            if constexpr (IS % DSTRIDE == DSTRIDE / 2 && IS / DSTRIDE < K::NPW) dma_piece(MIC<IS / DSTRIDE>{});
            // GELU of the lower half of chunk t+1: not in slots 0, 1 (the chain's last MFMAs are asm: nothing pads their results)
            if constexpr (NEXT) gelu_slot(Hn, Gn, bs0, MIC<0>{}, I_, MIC<2>{}, MIC<K::G2>{}, MIC<0>{});
            MF_SB;
        });
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bs1[0]), "+v"(bs1[1]));       // (the fragment stream has drained; this closes the two bias reads)
        MF_T(2)
    };
    {
        using T_ = std::true_type;
        using F_ = std::false_type;
        int t = 0;
        for (; t + 2 < nch; t += 2) {
            step(Ha, Hb, Ga, Gb, t, T_{});
            step(Hb, Ha, Gb, Ga, t + 1, T_{});
        }
        // nch is even (hidden % 64 == 0)
        step(Ha, Hb, Ga, Gb, t, T_{});
        step(Hb, Ha, Gb, Ga, t + 1, F_{});
    }

#ifdef MF_STAMP
    {       // diagnostic build: cycles per iteration in (wait + barrier, phase 1, phase 2) and the prologue's total, one row per wave; results invalid
        uint64_t t_end;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_end) :: "memory");
        if (lane == 0) {
            float* o = Out + ((int64_t)blockIdx.x * 4 + wave) * 8;
            for (int i = 0; i < 3; ++i) o[i] = (float)tacc[i] / nch;
            o[3] = (float)(t_loop0 - t_start);
            o[4] = (float)(t_end - t_loop0);
        }
        return;
    }
#endif
    // ---- epilogue: out = Y^T + b2 (Y^T started at x), fp32 -----------------------------------------------------------------------------
    // Stored lane-per-row a store instruction covers 32 rows x 32 bytes; instead a pass writes its accumulators into the wave's
    // staging area (padded rows: conflict-free), reads them back in memory order and stores whole 1-KiB runs.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the clamped LDS-DMA pieces of the last iterations)
    __builtin_amdgcn_s_barrier();                          // every wave is done with the rings
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    // (round 4 also had an epilogue form that wrote the NEXT block's norm1 from these accumulators -- measured neutral in the model,
    // taken out in round 5: tools/probes/block_tail_next_u.patch, profiles/r4_ln1_emit_in_model_ab.txt)
    {
        int row_c[3], col_c[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int q = 64 * c + lane;
            row_c[c] = q / K::PR;
            col_c[c] = q - row_c[c] * K::PR;
        }
        mf_for<0, TB * K::CP>([&](auto P_) {
            constexpr int tb = P_ / K::CP, cp = P_ % K::CP, NCB = CB / K::CP;
            char* wp = st_area + r * K::RS + 16 * h;
            mf_for<0, NCB * 4>([&](auto Q_) {
                constexpr int cbl = Q_ / 4, g = Q_ % 4, cb = cp * NCB + cbl;
                float4 a;
                mf_aget4<16 * (cb * TB + tb) + 4 * g>(a);
                *reinterpret_cast<float4*>(wp + (32 * cbl + 8 * g) * 4) = a;
            });
            float4 bb[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) bb[c] = *reinterpret_cast<const float4*>(B2 + cp * (C / K::CP) + 4 * col_c[c]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            mf_for<0, K::NI>([&](auto I_) {
                constexpr int i = I_, c = i % 3;
                const int row = row_c[c] + (i / 3) * K::RPI;
                float4 a = *reinterpret_cast<const float4*>(st_area + row * K::RS + 16 * col_c[c]);
                a.x += bb[c].x; a.y += bb[c].y; a.z += bb[c].z; a.w += bb[c].w;
                const int64_t t = tok0 + 32 * tb + row;
                if (t < M) *reinterpret_cast<float4*>(Out + t * C + cp * (C / K::CP) + 4 * col_c[c]) = a;
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the reads have left the area before the next pass writes it
        });
    }
}

// ---- weight packing ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int mf_swap23(int m) { return (m & ~12) | ((m & 4) << 1) | ((m & 8) >> 1); }

// one thread per 16-byte piece of the packed images
__global__ __launch_bounds__(256) void mlp_pack_kernel(const float* __restrict__ W1, const float* __restrict__ gamma, const float* __restrict__ W2,
                                                       char* __restrict__ out, int C, int hidden) {
    const int WU = 64 * C, nch = hidden / 32;
    const int64_t piece = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int ppc = 2 * WU / 16;              // pieces per chunk
    if (piece >= (int64_t)nch * ppc) return;
    const int c = (int)(piece / ppc), pi = (int)(piece % ppc);
    float v[8];
    if (pi < WU / 16) {                        // W1 image: slab q | row m | position   (C = 96: the last slab is 64 B wide)
        const int nfs = C / 64;
        int m, k0;
        if (pi < nfs * 256) {
            const int q = pi / 256, pos = pi % 8;
            m = (pi % 256) / 8;
            k0 = 64 * q + 8 * (pos ^ ((m >> 1) & 7));
        } else {
            const int p1 = pi - nfs * 256, pos = p1 % 4;
            m = p1 / 4;
            k0 = 64 * nfs + 8 * (pos ^ ((m >> 2) & 3));
        }
        const int hid = 32 * c + mf_swap23(m);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = W1[(int64_t)hid * C + k0 + e] * gamma[k0 + e];
    } else {                                   // W2 image: row ch | position
        const int p2 = pi - WU / 16, ch = p2 / 4, pos = p2 % 4;
        const int logical = pos ^ ((ch >> 2) & 3);
        const int hid0 = 32 * c + 8 * logical;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = W2[(int64_t)ch * hidden + hid0 + e];
    }
    uint4 u = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
    *reinterpret_cast<uint4*>(out + piece * 16) = u;
}
// b1'[hid] = b1[hid] + sum_k W1[hid][k] beta[k]   (fp32; one wave per hidden unit)
__global__ __launch_bounds__(256) void mlp_pack_bias_kernel(const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ beta,
                                                            float* __restrict__ out, int C, int hidden) {
    const int hid = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (hid >= hidden) return;
    float s = 0.f;
    for (int k = lane; k < C; k += 64) s = fmaf(W1[(int64_t)hid * C + k], beta[k], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) out[hid] = b1[hid] + s;
}

// proj chunk images (W1 format, rows in natural order: MFMA row m of chunk c = output channel 32c + m; no LayerNorm affine: the
// operand is the attention output), one thread per 16-byte piece
__global__ __launch_bounds__(256) void mlp_pack_proj_kernel(const float* __restrict__ Wp, char* __restrict__ out, int C) {
    const int WU = 64 * C, npc = C / 32;
    const int64_t piece = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int ppc = WU / 16;
    if (piece >= (int64_t)npc * ppc) return;
    const int c = (int)(piece / ppc), pi = (int)(piece % ppc), nfs = C / 64;
    int m, k0;
    if (pi < nfs * 256) {
        const int q = pi / 256, pos = pi % 8;
        m = (pi % 256) / 8;
        k0 = 64 * q + 8 * (pos ^ ((m >> 1) & 7));
    } else {
        const int p1 = pi - nfs * 256, pos = p1 % 4;
        m = p1 / 4;
        k0 = 64 * nfs + 8 * (pos ^ ((m >> 2) & 3));
    }
    const float* src = Wp + (int64_t)(32 * c + m) * C + k0;
    uint4 u = make_uint4(pack_bf16x2(src[0], src[1]), pack_bf16x2(src[2], src[3]), pack_bf16x2(src[4], src[5]), pack_bf16x2(src[6], src[7]));
    *reinterpret_cast<uint4*>(out + piece * 16) = u;
}
__global__ __launch_bounds__(256) void mlp_copy_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

static bool mf_shape_ok(int C, int hidden) { return (C == 96 || C == 192 || C == 384) && hidden == 4 * C; }
static int64_t mf_mlp_bytes(int C, int hidden) { return (int64_t)(hidden / 32) * 128 * C + (4ll * hidden + 1023) / 1024 * 1024; }

extern "C" int64_t mvit_mlp_fused_pack_bytes(int C, int hidden) {
    if (!mf_shape_ok(C, hidden)) return 0;
    return mf_mlp_bytes(C, hidden);      // b1' padded to whole KiB (it is brought in by 1-KiB LDS-DMA pieces)
}
extern "C" int64_t mvit_block_tail_pack_bytes(int C, int hidden) {
    if (!mf_shape_ok(C, hidden)) return 0;
    return (int64_t)(C / 32) * 64 * C + mf_mlp_bytes(C, hidden) + (4ll * C + 1023) / 1024 * 1024;       // proj chunks | MLP image | proj bias
}

extern "C" int mvit_mlp_fused_pack(const float* w1, const float* b1, const float* gamma, const float* beta, const float* w2, void* packed,
                                   int C, int hidden, void* stream) {
    if (!w1 || !b1 || !gamma || !beta || !w2 || !packed) return MVIT_EINVAL;
    if (!mf_shape_ok(C, hidden)) return MVIT_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
    const int64_t pieces = (int64_t)(hidden / 32) * 128 * C / 16;
    hipLaunchKernelGGL(mlp_pack_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, st, w1, gamma, w2, (char*)packed, C, hidden);
    MVIT_LAUNCH_CHECK();
    hipLaunchKernelGGL(mlp_pack_bias_kernel, dim3((unsigned)((hidden + 3) / 4)), dim3(256), 0, st, w1, b1, beta,
                       reinterpret_cast<float*>((char*)packed + (int64_t)(hidden / 32) * 128 * C), C, hidden);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}
extern "C" int mvit_block_tail_pack(const float* wproj, const float* bproj, const float* w1, const float* b1, const float* gamma,
                                    const float* beta, const float* w2, void* packed, int C, int hidden, void* stream) {
    if (!wproj || !bproj || !packed) return MVIT_EINVAL;
    if (!mf_shape_ok(C, hidden)) return MVIT_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
    char* base = reinterpret_cast<char*>(packed);
    const int64_t proj_bytes = (int64_t)(C / 32) * 64 * C, pieces = proj_bytes / 16;
    hipLaunchKernelGGL(mlp_pack_proj_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, st, wproj, base, C);
    MVIT_LAUNCH_CHECK();
    const int rc = mvit_mlp_fused_pack(w1, b1, gamma, beta, w2, base + proj_bytes, C, hidden, stream);
    if (rc != MVIT_OK) return rc;
    hipLaunchKernelGGL(mlp_copy_f32_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, st, bproj,
                       reinterpret_cast<float*>(base + proj_bytes + mf_mlp_bytes(C, hidden)), C);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}

template <int CB, int TB, bool PROJ>
static int mf_launch(const float* x, const void* o, const void* packed, const float* b2, float* out, int64_t M, int hidden, float eps, hipStream_t st) {
    using K = MfCfg<CB, TB>;
    const int ring = 6 * K::WU + K::B1BYTES + (PROJ ? K::BPBYTES : 0), stage = 4 * K::ST_PITCH + 12 * K::C;       // rings + biases | staging areas (+ b2, gamma, beta of the emitted LayerNorm)
    const int smem = ring > stage ? ring : stage;
    static DevFlags attr_tab;
    DevFlag attr_done = dev_flag(attr_tab);
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fused_kernel<CB, TB, PROJ>), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return MVIT_ELAUNCH;
        attr_done = true;
    }
    const int64_t tiles = (M + 128 * TB - 1) / (128 * TB);
    if (tiles >= (1ll << 31)) return MVIT_EUNSUPPORTED;
    const int nch = hidden / 32;
    const char* wpk = reinterpret_cast<const char*>(packed);
    const float* b1p = reinterpret_cast<const float*>(wpk + (PROJ ? (int64_t)CB * K::WU : 0) + (int64_t)nch * K::CHB);
    hipLaunchKernelGGL((mlp_fused_kernel<CB, TB, PROJ>), dim3((unsigned)tiles), dim3(256), smem, st, x, (const bf16_t*)o, wpk, b1p, b2, out, M, nch, eps);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}

extern "C" int mvit_mlp_fused_fwd(const float* x, const void* packed, const float* b2, float* out, int64_t M, int C, int hidden, float eps,
                                  int act_dtype, void* stream) {
    if (!x || !packed || !b2 || !out || M < 0) return MVIT_EINVAL;
    if (act_dtype != MVIT_BF16) return MVIT_EUNSUPPORTED;        // the fp32 path keeps LayerNorm + two GEMM launches
    if (!mf_shape_ok(C, hidden)) return MVIT_EUNSUPPORTED;
    if (M == 0) return MVIT_OK;
    hipStream_t st = as_stream(stream);
    switch (C) {
        case 384: return mf_launch<12, 1, false>(x, nullptr, packed, b2, out, M, hidden, eps, st);
        case 192: return mf_launch<6, 2, false>(x, nullptr, packed, b2, out, M, hidden, eps, st);
        default: return mf_launch<3, 2, false>(x, nullptr, packed, b2, out, M, hidden, eps, st);
    }
}
extern "C" int mvit_block_tail_fwd(const void* o, const float* resid, const void* packed, const float* b2, float* out, int64_t M, int C,
                                   int hidden, float eps, int act_dtype, void* stream) {
    if (!o || !resid || !packed || !b2 || !out || M < 0) return MVIT_EINVAL;
    if (act_dtype != MVIT_BF16) return MVIT_EUNSUPPORTED;
    if (!mf_shape_ok(C, hidden)) return MVIT_EUNSUPPORTED;
    if (M == 0) return MVIT_OK;
    hipStream_t st = as_stream(stream);
    switch (C) {
        case 384: return mf_launch<12, 1, true>(resid, o, packed, b2, out, M, hidden, eps, st);
        case 192: return mf_launch<6, 2, true>(resid, o, packed, b2, out, M, hidden, eps, st);
        default: return mf_launch<3, 2, true>(resid, o, packed, b2, out, M, hidden, eps, st);
    }
}
