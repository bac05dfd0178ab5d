// Pooling conv (depthwise 3x3x3, stride (1,S,S), zero pad 1, one 96-channel kernel shared by all heads) + LayerNorm(96),
// spatial strides 1 and 2 (98 % of the pooled tokens): the "march" kernels -- forward (inference / training), the stride-1 data
// gradient, and (at the end of the file) the weight gradient with the LayerNorm backward optionally fused in front.
//   reference: attention_pool, slowfast/models/attention.py:12-83 (conv variant) with norm_{q,k,v} of :185,199,213.
//
// fp32-VALU-bound op (27 taps x 96 channels per token against 192 B in + 192 B out at stride 1; DESIGN.md section 8b has the
// counters and ablations).  Structure:
//   * workgroup = 7 waves = a 7 x 7 output tile of one (batch, head); wave = one output row, lane = one channel PAIR (48 of
//     the 64 lanes carry data; every token grid of the model -- 112, 56, 28, 14, 7 -- is a multiple of 7, so no tile is partial);
//     blockIdx.y >= set_bh selects a SECOND tensor (the k / v pair of a block: adjacent head groups of the fused qkv buffer, own
//     conv weights and LayerNorm parameters, outputs back to back) -- one launch of 2 x B x heads x tiles workgroups;
//   * the workgroup marches over the T input frames; a frame's halo tile (IH x IW tokens x 96 channels) arrives in LDS by
//     global_load_lds (inline asm: SGPR frame base + 32-bit lane offset), double-buffered so the next frame is in flight under
//     the current frame's arithmetic; halo cells outside the image are zeroed once and never written again;
//   * three rolling accumulator sets (output frames f-1, f, f+1) with COMPILE-TIME names: the frame body is instantiated for
//     the three phases f mod 3, nothing is copied when the window moves; each LDS value is read once per dy and used for 9
//     taps, arithmetic in plain v_fma_f32 (fp16 build: the stride-2 form lets the compiler fold the conversions into v_fma_mix_f32, the
//     stride-1 form keeps them apart -- unfold_conversions below), the 27 weight
//     reads per input row stay LDS reads (common.h::lds_opaque -- laundering the generic pointer made them FLAT loads);
//   * the LayerNorm of a finished frame runs IN REGISTERS: the 96 channels of a token are the 48 lanes of one wave; the sums
//     of a frame's 7 tokens are folded into each other (wave_sum_rows: v_permlane32_swap, v_permlane16_swap, 4 DPP levels on two
//     registers) and the per-token scalars fetched by v_readlane -- no LDS stage, no barrier; one s_barrier per frame in total.
// MODE 0: forward.  MODE 1: training forward, also writes xhat = (conv - mean) * rstd and rstd (what the LayerNorm backward
// needs).  MODE 2 ("plain", stride 1): the bare convolution with mirrored taps written token-major into a channel slice of a
// [B][tokens][out_ld] buffer = the DATA gradient of the stride-1 pooling conv (attention.py:56 backward).
#include <type_traits>

#include "common.h"

typedef __attribute__((ext_vector_type(2))) float f32x2;

// timing ablations (tools/r5_march_abl.sh builds variant libraries; the product is always built with 0):
//   1 no tap arithmetic   2 no LayerNorm (pack the raw conv)   4 no DMA after frame 0   8 no output stores   16 no per-frame barrier
#ifndef MARCH_ABL
#define MARCH_ABL 0
#endif

template <typename TA, int S>
struct March {
    static constexpr int ROWS = 7, XO = 7;
    static constexpr int IH = S * (ROWS - 1) + 3, IW = S * (XO - 1) + 3;
    static constexpr int NT = 64 * ROWS;
    static constexpr int CW = 16 / (int)sizeof(TA), CPT = 96 / CW;      // channels per 16-byte chunk, chunks per token
    static constexpr int NCHUNK = IH * IW * CPT;
    static constexpr int PF = (NCHUNK + NT - 1) / NT;                  // DMA instructions per thread and frame
    static constexpr int IN_BYTES = IH * IW * 96 * (int)sizeof(TA);
    static constexpr int NBUF = (2 * IN_BYTES + 27 * 96 * 4 <= 160 * 1024) ? 2 : 1;   // (fp32, stride 2: one 86 KiB tile)
    static constexpr int W_BYTES = 27 * 96 * 4;
    static constexpr int SMEM = NBUF * IN_BYTES + W_BYTES;
};

// Totals over the 64 lanes of a wave of N <= 8 independent per-lane values, TRANSPOSED: instead of one 6-level butterfly per value
// (8 VALU each: 64 for 8 values) the values are folded into each other -- two registers become one whose lane halves carry the
// half-sums of the two (v_permlane32_swap + add), two of those become one whose four 16-lane rows carry four values
// (v_permlane16_swap + add), and only the within-row levels (4 DPP adds) run per register: 6 + 3 + 2 x 4 = 20 VALU for 8 values.
// On return q[m] holds, in every lane of row r, the total of value 4m + ROW_TO_VAL[r] (ROW_TO_VAL = 0, 2, 1, 3): the caller
// finishes its per-token arithmetic on the two registers (each lane = one token's scalar) and fetches scalars with
// wave_sum_pick (v_readlane: a wave-uniform SGPR operand, no broadcast arithmetic).
template <int N>
__device__ __forceinline__ void wave_sum_rows(const float (&v)[N], float (&q)[2]) {
    static_assert(N <= 8, "at most 8 values");
    float h[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float a = 2 * k < N ? v[2 * k] : 0.f, b = 2 * k + 1 < N ? v[2 * k + 1] : 0.f;
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
        h[k] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);      // lanes 0-31: half-sums of a, lanes 32-63: of b
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(h[2 * m]), __float_as_uint(h[2 * m + 1]), false, false);
        q[m] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);      // rows: v[4m], v[4m+2], v[4m+1], v[4m+3]
    }
#define DPP_LEVEL(CTRL)                                                                                           \
    _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                                 \
        q[m] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(q[m]), CTRL, 0xF, 0xF, false));
    DPP_LEVEL(0xB1)      // quad_perm [1,0,3,2]
    DPP_LEVEL(0x4E)      // quad_perm [2,3,0,1]
    DPP_LEVEL(0x141)     // row_half_mirror
    DPP_LEVEL(0x140)     // row_mirror
#undef DPP_LEVEL
}
// the scalar of value i from the row registers of wave_sum_rows (wave-uniform)
__device__ __forceinline__ float wave_sum_pick(const float (&q)[2], int i) {
    const int r = i & 3, row = r == 0 ? 0 : (r == 2 ? 1 : (r == 1 ? 2 : 3));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q[i >> 2]), 16 * row));
}

template <typename TA>
__device__ __forceinline__ f32x2 lds_pair(const char* p);
template <>
__device__ __forceinline__ f32x2 lds_pair<bf16_t>(const char* p) {
    const uint32_t u = *reinterpret_cast<const uint32_t*>(p);
    f32x2 r;
    r.x = lo16_to_f32(u);
    r.y = hi16_to_f32(u);
    return r;
}
// fp16 build, stride 1: keep the conversions as instructions of their own.  Left alone the compiler folds them into the taps as v_fma_mix_f32
// (fewer instructions), which issue below the plain v_fmac_f32 rate on gfx950 like the other packed-encoding fp32 arithmetic: the stride-1
// kernel ran 22 % behind the bf16 build inside the fp16 forward (profiles/r5_fwd_fp16_vs_bf16_per_kernel.txt); with explicit conversions 48.5 -> 45.0 us
// at stage 3 alone, 187.6 -> 178 at block 0.  The stride-2 form has fewer taps per loaded value and is better off folded (37.3 vs 41.5 us):
// profiles/r5_pool_f16_fma_mix_ab.txt.
template <int S>
__device__ __forceinline__ void unfold_conversions(f32x2& r) {
#if defined(MVIT_HALF_IS_FP16) && !defined(MARCH_FMA_MIX)      // (MARCH_FMA_MIX: A/B builds of the folded form)
    if constexpr (S == 1) asm volatile("" : "+v"(r.x), "+v"(r.y));
#endif
}
template <>
__device__ __forceinline__ f32x2 lds_pair<float>(const char* p) {
    return *reinterpret_cast<const f32x2*>(p);
}
template <typename TA>
__device__ __forceinline__ void st_pair(TA* p, float a, float b);
template <>
__device__ __forceinline__ void st_pair<bf16_t>(bf16_t* p, float a, float b) { *reinterpret_cast<uint32_t*>(p) = pack_bf16x2(a, b); }
template <>
__device__ __forceinline__ void st_pair<float>(float* p, float a, float b) { *reinterpret_cast<float2*>(p) = make_float2(a, b); }

template <typename TA, int S, int MODE>
__global__ __launch_bounds__(448) void pool_march_kernel(const TA* __restrict__ in, int64_t ld, int chan_off,
                                                         const float* __restrict__ w, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, TA* __restrict__ out,
                                                         TA* __restrict__ xhat, float* __restrict__ rstd_out, int heads, int T,
                                                         int H, int W, int Ho, int Wo, float eps, int64_t out_ld,
                                                         int out_chan_off, int out_heads, int set_bh, const float* __restrict__ w2,
                                                         const float* __restrict__ gamma2, const float* __restrict__ beta2) {
    // set_bh > 0: TWO tensors in one launch (the k and the v pooling conv of a block: adjacent head groups of the fused qkv buffer,
    // own conv weights / LayerNorm parameters, outputs back to back): blockIdx.y = set * set_bh + (b * heads + g)
    using P = March<TA, S>;
    constexpr int ES = (int)sizeof(TA);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* wl = reinterpret_cast<float*>(smem + P::NBUF * P::IN_BYTES);        // [tap][channel] fp32
    const int tid = threadIdx.x;
    const int row = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const bool live = lane < 48;
    const int cp = live ? lane : 47;                                           // channel pair (dummy lanes shadow pair 47)
    const int tiles_x = (Wo + P::XO - 1) / P::XO;
    const int tx0 = (blockIdx.x % tiles_x) * P::XO, ty0 = (blockIdx.x / tiles_x) * P::ROWS;
    const int bh = blockIdx.y;
    int bhs = bh, hoff = 0;
    if (set_bh > 0 && bh >= set_bh) { bhs = bh - set_bh; hoff = heads; w = w2; gamma = gamma2; beta = beta2; }
    const int b = bhs / heads, g = bhs - b * heads;
    const int64_t Nin = (int64_t)T * H * W;
    const char* base = reinterpret_cast<const char*>(in + (int64_t)b * Nin * ld + chan_off + (hoff + g) * 96);
    const int y_in0 = S * ty0 - 1, x_in0 = S * tx0 - 1;

    // ---- one-time setup: weights, zeroed tiles, per-thread DMA offsets ------------------------------------------------------
    for (int i = tid; i < 27 * 96; i += P::NT) {
        const int tap = i / 96, c = i - tap * 96;
        wl[i] = w[c * 27 + (MODE == 2 ? 26 - tap : tap)];
    }
    for (int i = tid; i < P::NBUF * P::IN_BYTES / 16; i += P::NT) *reinterpret_cast<uint4*>(smem + i * 16) = make_uint4(0, 0, 0, 0);
    uint32_t poff[P::PF];       // byte offset of this thread's chunk inside one frame; 0xffffffff = outside the image / unused
#pragma unroll
    for (int i = 0; i < P::PF; ++i) {
        const int c = tid + P::NT * i;
        poff[i] = 0xffffffffu;
        if (c < P::NCHUNK) {
            const int tok = c / P::CPT, ch = c - tok * P::CPT;
            const int iy = tok / P::IW, ix = tok - iy * P::IW;
            const int y = y_in0 + iy, x = x_in0 + ix;
            if (y >= 0 && y < H && x >= 0 && x < W) poff[i] = (uint32_t)(((int64_t)(y * W + x) * ld) * ES + ch * 16);
        }
    }
    const int64_t frame_bytes = (int64_t)H * W * ld * ES;
    const uint32_t smem_a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    auto dma = [&](int f, int buf) {
        const char* fb = base + f * frame_bytes;                                // wave-uniform
#pragma unroll
        for (int i = 0; i < P::PF; ++i) {
            const uint32_t dst = __builtin_amdgcn_readfirstlane(smem_a + buf * P::IN_BYTES + 1024 * (row + P::ROWS * i));
            if (poff[i] != 0xffffffffu)      // lanes outside the image stay masked: their LDS bytes keep the zeros written above
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(poff[i]), "s"(fb) : "memory");
        }
    };
    __syncthreads();                     // zeros and weights are in place before the first DMA piece may land
    dma(0, 0);

    const int yo = ty0 + row;
    const bool row_ok = yo < Ho;
    f32x2 g2 = {0.f, 0.f}, b2 = {0.f, 0.f};
    if (MODE != 2) {
        g2.x = gamma[2 * cp]; g2.y = gamma[2 * cp + 1];
        b2.x = beta[2 * cp]; b2.y = beta[2 * cp + 1];
    }

    f32x2 acc[3][P::XO];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int x = 0; x < P::XO; ++x) acc[a][x] = f32x2{0.f, 0.f};

    // ---- finished output frame (accumulator set 0): LayerNorm in registers now, stores one frame LATER ----------------------
    // The results wait in registers (packed to the output type) and are stored at the top of the next frame, behind that
    // frame's DMA request and in front of its arithmetic: when the loop comes round to `s_waitcnt vmcnt(0)` again, both the DMA
    // and these stores are a whole frame of arithmetic old.  (Stored right after the LayerNorm, every frame waited for its own
    // store acknowledgements at that vmcnt(0) -- CDNA4 counts stores in vmcnt -- with the whole workgroup behind the barrier.)
    typedef typename std::conditional<sizeof(TA) == 2, uint32_t, float2>::type pair_t;
    pair_t pend_o[P::XO], pend_h[MODE == 1 ? P::XO : 1];
    float pend_r = 0.f;
    int pend_fo = -1;
    auto pack = [&](float a, float b2_) -> pair_t {
        if constexpr (sizeof(TA) == 2) return pack_bf16x2(a, b2_);
        else return make_float2(a, b2_);
    };
    auto finalize = [&](auto SET, int fo) {        // arithmetic only; SET = which accumulator set holds output frame fo
        constexpr int a0 = decltype(SET)::value;
        pend_fo = fo;
        if (!row_ok) return;             // wave-uniform
        if constexpr (MODE == 2 || (MARCH_ABL & 2)) {
#pragma unroll
            for (int x = 0; x < P::XO; ++x) pend_o[x] = pack(acc[a0][x].x, acc[a0][x].y);
        } else {
            float part[P::XO], q[2];
#pragma unroll
            for (int x = 0; x < P::XO; ++x) part[x] = live ? acc[a0][x].x + acc[a0][x].y : 0.f;
            wave_sum_rows<P::XO>(part, q);
            q[0] *= (1.0f / 96.0f);
            q[1] *= (1.0f / 96.0f);
            f32x2 d[P::XO];
#pragma unroll
            for (int x = 0; x < P::XO; ++x) {
                const float mean = wave_sum_pick(q, x);
                d[x].x = acc[a0][x].x - mean;
                d[x].y = acc[a0][x].y - mean;
                part[x] = live ? d[x].x * d[x].x + d[x].y * d[x].y : 0.f;
            }
            wave_sum_rows<P::XO>(part, q);
            // v_rsq_f32 (1 ulp): one instruction, not the 25-instruction IEEE sqrt + divide -- and on the two row registers (every
            // lane of a row = that row's token), not once per token
            q[0] = __builtin_amdgcn_rsqf(q[0] * (1.0f / 96.0f) + eps);
            q[1] = __builtin_amdgcn_rsqf(q[1] * (1.0f / 96.0f) + eps);
#pragma unroll
            for (int x = 0; x < P::XO; ++x) {
                const float rstd = wave_sum_pick(q, x);
                const float h0 = d[x].x * rstd, h1 = d[x].y * rstd;
                if constexpr (MODE == 1) {
                    pend_h[x] = pack(h0, h1);
                    pend_r = (x == 0 || lane == x) ? rstd : pend_r;
                }
                pend_o[x] = pack(fmaf(h0, g2.x, b2.x), fmaf(h1, g2.y, b2.y));
            }
        }
    };
    auto flush = [&]() {                 // stores of the pending frame
        if (pend_fo < 0 || !row_ok || (MARCH_ABL & 8)) return;
        const int fo = pend_fo;
        if constexpr (MODE == 2) {
            const int ob = bh / out_heads, og = bh - ob * out_heads;
            TA* o = out + ((int64_t)ob * T * Ho * Wo + ((int64_t)fo * Ho + yo) * Wo + tx0) * out_ld + out_chan_off + og * 96 + 2 * cp;
#pragma unroll
            for (int x = 0; x < P::XO; ++x)
                if (tx0 + x < Wo && live) *reinterpret_cast<pair_t*>(o + x * out_ld) = pend_o[x];
        } else {
            const int64_t orow = (((int64_t)bh * T + fo) * Ho + yo) * Wo + tx0;     // token index of x = 0
            TA* o = out + orow * 96 + 2 * cp;
#pragma unroll
            for (int x = 0; x < P::XO; ++x)
                if (tx0 + x < Wo && live) {
                    if constexpr (MODE == 1) *reinterpret_cast<pair_t*>(xhat + (orow + x) * 96 + 2 * cp) = pend_h[x];
                    *reinterpret_cast<pair_t*>(o + x * 96) = pend_o[x];
                }
            if constexpr (MODE == 1)
                if (lane < P::XO && tx0 + lane < Wo) rstd_out[orow + lane] = pend_r;
        }
    };

    // ---- the march ---------------------------------------------------------------------------------------------------------
    // Output frame fo lives in accumulator set fo % 3 for its whole life (the frame body is instantiated for the three phases
    // f % 3, so the set indices are compile-time: no register copies rotate the sets -- 28 v_mov per frame before round 5).
    auto frame = [&](auto PH, int f) {
        constexpr int ph = decltype(PH)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // frame f has landed (this wave's pieces) ...
        if (!(MARCH_ABL & 16)) __builtin_amdgcn_s_barrier();       // ... and everyone's; all waves are past frame f-1's reads
        const int cur = P::NBUF == 2 ? (f & 1) : 0;
        if (P::NBUF == 2 && f + 1 < T && !(MARCH_ABL & 4)) dma(f + 1, cur ^ 1);
        flush();                           // output frame f-2
        const char* tile = smem + cur * P::IN_BYTES + cp * 2 * ES;
        // keep the 27 weight reads inside the loop (hoisted they would pin 54 registers) -- as LDS reads (see lds_opaque)
        const lds_cptr_t wm = lds_opaque(reinterpret_cast<const char*>(wl) + cp * 8);
#pragma unroll 1
        for (int dy = 0; dy < ((MARCH_ABL & 1) ? 0 : 3); ++dy) {
            f32x2 xin[P::IW];
            const char* rp = tile + (S * row + dy) * P::IW * 96 * ES;
#pragma unroll
            for (int ix = 0; ix < P::IW; ++ix) {
                xin[ix] = lds_pair<TA>(rp + ix * 96 * ES);
                if constexpr (sizeof(TA) == 2) unfold_conversions<S>(xin[ix]);
            }
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {    // input frame f is tap dt of output frame f + 1 - dt -> set (f + 1 - dt) % 3
                const int a = (ph + 4 - dt) % 3;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const f32x2 wv = lds_ld<f32x2>(wm, ((dt * 3 + dy) * 3 + dx) * 384);
#pragma unroll
                    for (int x = 0; x < P::XO; ++x) {      // plain v_fma_f32 x 2: v_pk_fma_f32 issues at well under half their rate on gfx950
                        acc[a][x].x = fmaf(wv.x, xin[S * x + dx].x, acc[a][x].x);
                        acc[a][x].y = fmaf(wv.y, xin[S * x + dx].y, acc[a][x].y);
                    }
                }
            }
        }
        if (P::NBUF == 1 && f + 1 < T) {   // single buffer: the next frame may only be requested once every wave has read this one
            __builtin_amdgcn_s_barrier();
            dma(f + 1, 0);
        }
        constexpr int done = (ph + 2) % 3;  // output frame f-1 is complete; its set then restarts as output frame f+2
        if (f >= 1) finalize(std::integral_constant<int, done>{}, f - 1);
#pragma unroll
        for (int x = 0; x < P::XO; ++x) acc[done][x] = f32x2{0.f, 0.f};
    };
    for (int f0 = 0; f0 < T; f0 += 3) {
        frame(std::integral_constant<int, 0>{}, f0);
        if (f0 + 1 < T) frame(std::integral_constant<int, 1>{}, f0 + 1);
        if (f0 + 2 < T) frame(std::integral_constant<int, 2>{}, f0 + 2);
    }
    flush();                               // output frame T-2
    switch ((T - 1) % 3) {
        case 0: finalize(std::integral_constant<int, 0>{}, T - 1); break;
        case 1: finalize(std::integral_constant<int, 1>{}, T - 1); break;
        default: finalize(std::integral_constant<int, 2>{}, T - 1); break;
    }
    flush();
}

template <typename TA, int S, int MODE>
static int launch_march(const void* in, int64_t ld, int chan_off, const float* w, const float* gamma, const float* beta, void* out,
                        void* xhat, float* rstd, int B, int heads, int T, int H, int W, int Ho, int Wo, float eps, int64_t out_ld,
                        int out_chan_off, int out_heads, hipStream_t st, const float* w2 = nullptr, const float* gamma2 = nullptr,
                        const float* beta2 = nullptr) {
    using P = March<TA, S>;
    dim3 grid(((Wo + P::XO - 1) / P::XO) * ((Ho + P::ROWS - 1) / P::ROWS), B * heads * (w2 ? 2 : 1));
    static DevFlags attr_done_tab; DevFlag attr_done = dev_flag(attr_done_tab);
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pool_march_kernel<TA, S, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                P::SMEM) != hipSuccess)
            return MVIT_ELAUNCH;
        attr_done = true;
    }
    hipLaunchKernelGGL((pool_march_kernel<TA, S, MODE>), grid, dim3(P::NT), P::SMEM, st, (const TA*)in, ld, chan_off, w, gamma, beta,
                       (TA*)out, (TA*)xhat, rstd, heads, T, H, W, Ho, Wo, eps, out_ld, out_chan_off, out_heads, w2 ? B * heads : 0, w2, gamma2, beta2);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}

// internal entry points (pool.hip / pool_bwd.hip): forward (+ saved statistics) for strides 1 and 2 ...
int mvit_internal_pool_march_fwd(const void* qkv, int64_t ld, int chan_off, const float* w, const float* gamma, const float* beta, void* out,
                                 void* xhat, float* rstd, int B, int heads, int T, int H, int W, int stride_hw, float eps, int act_dtype,
                                 hipStream_t st, const float* w2, const float* gamma2, const float* beta2) {
    // (w2 / gamma2 / beta2 non-null: the pair form -- a second tensor's head group follows the first's in the input, see the kernel)
    const int Ho = (H - 1) / stride_hw + 1, Wo = (W - 1) / stride_hw + 1;
    // frame / tile offsets are 32-bit byte offsets from a per-(batch, head) 64-bit base
    if ((int64_t)H * W * ld * 4 >= (1ll << 31)) return MVIT_EUNSUPPORTED;
#define MARCH(TA, S)                                                                                                                   \
    (xhat ? launch_march<TA, S, 1>(qkv, ld, chan_off, w, gamma, beta, out, xhat, rstd, B, heads, T, H, W, Ho, Wo, eps, 0, 0, 1, st, w2, gamma2, beta2)    \
          : launch_march<TA, S, 0>(qkv, ld, chan_off, w, gamma, beta, out, nullptr, nullptr, B, heads, T, H, W, Ho, Wo, eps, 0, 0, 1, st, w2, gamma2, beta2))
    if (act_dtype == MVIT_BF16) return stride_hw == 1 ? MARCH(bf16_t, 1) : MARCH(bf16_t, 2);
    return stride_hw == 1 ? MARCH(float, 1) : MARCH(float, 2);
#undef MARCH
}

// ... and the data gradient of the stride-1 conv: dconv [B*heads][T*H*W][96] -> the (chan_off) slice of dqkv [B][T*H*W][ld]
int mvit_internal_pool_march_dgrad1(const void* dconv, const float* w, void* dqkv, int64_t ld, int chan_off, int B, int heads, int T,
                                    int H, int W, int act_dtype, hipStream_t st) {
    if ((int64_t)H * W * 96 * 4 >= (1ll << 31)) return MVIT_EUNSUPPORTED;
    if (act_dtype == MVIT_BF16)
        return launch_march<bf16_t, 1, 2>(dconv, 96, 0, w, nullptr, nullptr, dqkv, nullptr, nullptr, B * heads, 1, T, H, W, H, W, 0.f, ld,
                                          chan_off, heads, st);
    return launch_march<float, 1, 2>(dconv, 96, 0, w, nullptr, nullptr, dqkv, nullptr, nullptr, B * heads, 1, T, H, W, H, W, 0.f, ld,
                                     chan_off, heads, st);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Weight gradient of the pooling conv in the march form (round 5; 16-bit builds, strides 1 and 2):
//   dw[c][dt][dy][dx] = sum_tokens d_conv[(to, yo, xo)][c] * in[(to + dt - 1, S yo + dy - 1, S xo + dx - 1)][c]      (attention.py:56 backward)
// Same skeleton as the forward (7 waves = the 7 rows of a 7 x 7 d_conv tile, lane = channel pair, input halo frames by LDS-DMA,
// double-buffered).  What is different:
//   * the sum over a row's 7 tokens is a DOT PRODUCT of two 16-bit activation vectors, so it runs on v_dot2c_f32_{bf16,f16}
//     (two multiply-adds per instruction, fp32 accumulation, operands used as they are stored -- no unpacking and, unlike in the
//     forward, no weight to round): the channel-pair registers of two neighbouring tokens are turned into token-pair registers of
//     one channel by v_perm_b32, a row's 7 tokens become 4 pairs (the last one padded with a zero), and a tap costs 4 dot2
//     instead of 7 fma: 216 + 62 permutes per frame against 378 + 68 unpack operations in the 8-wide form of pool.hip;
//   * a lane needs d_conv only for its own row and channel pair: plain global loads into registers, one frame ahead, kept as a
//     3-frame ring of token-pair registers whose slot names are compile-time (3-phase frame body) -- no LDS, no barrier for them;
//   * the 27 x 2 sums of a lane live in registers over all frames; at the end the 7 rows are added IN ROW ORDER through one
//     [tap][96] LDS slab (fixed summation order: bit-reproducible), one partial row [96 * 27] per workgroup for pool_reduce.
// 7 x 7 tiles cover every token grid of the model exactly (the 8-wide tiles of pool.hip waste 23 % at 28 x 28 and 14 x 14).
// ------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dot2_16(uint32_t a, uint32_t b, float c) {
#ifdef MVIT_HALF_IS_FP16
    typedef __attribute__((ext_vector_type(2))) _Float16 v2;
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(v2, a), __builtin_bit_cast(v2, b), c, false);
#else
    typedef __attribute__((ext_vector_type(2))) __bf16 v2;
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2, a), __builtin_bit_cast(v2, b), c, false);
#endif
}
// (lo16 of a, lo16 of b) and (hi16 of a, hi16 of b): the two channels of a lane's pair, each as a two-token vector
__device__ __forceinline__ uint32_t pair_lo(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }
__device__ __forceinline__ uint32_t pair_hi(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// LNB = true: the LayerNorm BACKWARD of the pooled tensor rides in front (round 5): instead of d_conv the kernel reads what the training
// forward saved (xhat, rstd) and the incoming gradient dout, forms d_conv = rstd (g dy - mean(g dy) - xhat mean(g dy xhat)) for its 7
// tokens per frame in registers (two transposed wave reductions), rounds it to the 16-bit type exactly as the separate pass did, WRITES
// it (the data-gradient kernel reads it next), uses it for the weight gradient, and keeps the per-lane sums of d_gamma = sum dy xhat and
// d_beta = sum dy, which leave as one [192] partial row per workgroup next to the [2592] one.  The row-wise pass of pool_bwd.hip
// (one more read of xhat / dout, one more write + read of d_conv, one more launch) disappears.
template <int S, bool LNB>
__global__ __launch_bounds__(448) void pool_wgrad_march_kernel(const bf16_t* __restrict__ in, int64_t ld, int chan_off,
                                                               const bf16_t* __restrict__ dconv, float* __restrict__ part, int heads,
                                                               int T, int H, int W, int Ho, int Wo, int set_bh,
                                                               const bf16_t* __restrict__ xhat, const bf16_t* __restrict__ dout,
                                                               const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                               const float* __restrict__ gamma2, bf16_t* __restrict__ dconv_out,
                                                               float* __restrict__ part_ln) {
    using P = March<bf16_t, S>;
    constexpr int ES = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int row = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const bool live = lane < 48;
    const int cp = live ? lane : 47;
    const int tiles_x = (Wo + P::XO - 1) / P::XO;
    const int tx0 = (blockIdx.x % tiles_x) * P::XO, ty0 = (blockIdx.x / tiles_x) * P::ROWS;
    const int bh = blockIdx.y;
    const int hoff = (set_bh > 0 && bh >= set_bh) ? heads : 0, bhs = hoff ? bh - set_bh : bh;
    const int b = bhs / heads, g = bhs - b * heads;
    const int64_t Nin = (int64_t)T * H * W;
    const char* base = reinterpret_cast<const char*>(in + (int64_t)b * Nin * ld + chan_off + (hoff + g) * 96);
    const int y_in0 = S * ty0 - 1, x_in0 = S * tx0 - 1;

    for (int i = tid; i < P::NBUF * P::IN_BYTES / 16; i += P::NT) *reinterpret_cast<uint4*>(smem + i * 16) = make_uint4(0, 0, 0, 0);
    uint32_t poff[P::PF];
#pragma unroll
    for (int i = 0; i < P::PF; ++i) {
        const int c = tid + P::NT * i;
        poff[i] = 0xffffffffu;
        if (c < P::NCHUNK) {
            const int tok = c / P::CPT, ch = c - tok * P::CPT;
            const int iy = tok / P::IW, ix = tok - iy * P::IW;
            const int y = y_in0 + iy, x = x_in0 + ix;
            if (y >= 0 && y < H && x >= 0 && x < W) poff[i] = (uint32_t)(((int64_t)(y * W + x) * ld) * ES + ch * 16);
        }
    }
    const int64_t frame_bytes = (int64_t)H * W * ld * ES;
    const uint32_t smem_a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    auto dma = [&](int f, int buf) {
        const char* fb = base + f * frame_bytes;
#pragma unroll
        for (int i = 0; i < P::PF; ++i) {
            const uint32_t dst = __builtin_amdgcn_readfirstlane(smem_a + buf * P::IN_BYTES + 1024 * (row + P::ROWS * i));
            if (poff[i] != 0xffffffffu)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(poff[i]), "s"(fb) : "memory");
        }
    };
    __syncthreads();
    dma(0, 0);

    // d_conv of this lane: row yo, tokens tx0 .. tx0+6, channel pair cp of output frame fo -> 7 dwords (zero outside the grid / for the
    // 16 idle lanes, whose sums therefore stay zero)
    const int yo = ty0 + row;
    const bool row_ok = yo < Ho;
    const int64_t tok0 = (int64_t)bh * T * Ho * Wo + (int64_t)(row_ok ? yo : 0) * Wo + tx0;      // token index of (frame 0, row yo, x = tx0)
    const uint32_t* drow = reinterpret_cast<const uint32_t*>((LNB ? dout : dconv) + tok0 * 96) + cp;
    const uint32_t* hrow = LNB ? reinterpret_cast<const uint32_t*>(xhat + tok0 * 96) + cp : nullptr;
    const int64_t dframe = (int64_t)Ho * Wo * 48;      // dwords per output frame
    uint32_t raw[P::XO], rawh[LNB ? P::XO : 1];
    float rs_l = 0.f;                                  // LNB: rstd of token x in lane x
    float lg0 = 0.f, lg1 = 0.f, dga0 = 0.f, dga1 = 0.f, dbe0 = 0.f, dbe1 = 0.f;
    if constexpr (LNB) {
        const float* gm = (set_bh > 0 && bh >= set_bh) ? gamma2 : gamma;
        lg0 = gm[2 * cp]; lg1 = gm[2 * cp + 1];
    }
    auto load_d = [&](int fo) {
        const uint32_t* p = drow + fo * dframe;
#pragma unroll
        for (int x = 0; x < P::XO; ++x) raw[x] = (tx0 + x < Wo) ? p[(tx0 + x < Wo ? x : 0) * 48] : 0u;      // (the address is clamped into the row: the compiler may hoist the load over the test)
        if constexpr (LNB) {
            const uint32_t* ph = hrow + fo * dframe;
#pragma unroll
            for (int x = 0; x < P::XO; ++x) rawh[x] = (tx0 + x < Wo) ? ph[(tx0 + x < Wo ? x : 0) * 48] : 0u;
            const int lx = lane < P::XO && tx0 + lane < Wo ? lane : 0;
            rs_l = rstd[tok0 + (int64_t)fo * Ho * Wo + lx];
        }
    };
    uint32_t pd[3][4][2];       // [output frame % 3][token pair][channel of the lane's pair]
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int k = 0; k < 4; ++k) pd[s][k][0] = pd[s][k][1] = 0u;
    const uint32_t keep = (live && row_ok) ? 0xffffffffu : 0u;
    auto convert = [&](auto SLOT, bool ok, int fo) {       // raw -> token-pair registers of slot SLOT (zeros for a frame outside [0, T))
        constexpr int s = decltype(SLOT)::value;
        const uint32_t m = ok ? keep : 0u;
        if constexpr (LNB) {
            if (ok) {                                      // wave-uniform: raw = dout, rawh = xhat of output frame fo -> raw = d_conv
                // (the unpacked values are formed twice -- once for the sums, once for the result -- rather than kept: 42 registers fewer,
                // which is what keeps two workgroups per CU at stride 1)
                float p1[P::XO], p2[P::XO], q1[2], q2[2];
                const float kf = keep ? 1.f : 0.f;
#pragma unroll
                for (int x = 0; x < P::XO; ++x) {
                    const float y0 = lo16_to_f32(raw[x]) * kf, y1 = hi16_to_f32(raw[x]) * kf;      // (tokens outside the grid loaded 0)
                    const float h0 = lo16_to_f32(rawh[x]), h1 = hi16_to_f32(rawh[x]);
                    dga0 = fmaf(y0, h0, dga0); dga1 = fmaf(y1, h1, dga1);
                    dbe0 += y0; dbe1 += y1;
                    const float g0 = y0 * lg0, g1 = y1 * lg1;
                    p1[x] = g0 + g1;
                    p2[x] = fmaf(g0, h0, g1 * h1);
                }
                wave_sum_rows<P::XO>(p1, q1);
                wave_sum_rows<P::XO>(p2, q2);
                q1[0] *= (1.0f / 96.0f); q1[1] *= (1.0f / 96.0f);
                q2[0] *= (1.0f / 96.0f); q2[1] *= (1.0f / 96.0f);
                bf16_t* dc = dconv_out + (tok0 + (int64_t)fo * Ho * Wo) * 96 + 2 * cp;
#pragma unroll
                for (int x = 0; x < P::XO; ++x) {
                    const float c1 = wave_sum_pick(q1, x), c2 = wave_sum_pick(q2, x);
                    const float r = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rs_l), x));
                    const float g0 = lo16_to_f32(raw[x]) * kf * lg0, g1 = hi16_to_f32(raw[x]) * kf * lg1;
                    const float h0 = lo16_to_f32(rawh[x]), h1 = hi16_to_f32(rawh[x]);
                    raw[x] = pack_bf16x2(r * (g0 - c1 - h0 * c2), r * (g1 - c1 - h1 * c2));
                    if (keep && tx0 + x < Wo) *reinterpret_cast<uint32_t*>(dc + x * 96) = raw[x];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            pd[s][k][0] = pair_lo(raw[2 * k], raw[2 * k + 1]) & m;
            pd[s][k][1] = pair_hi(raw[2 * k], raw[2 * k + 1]) & m;
        }
        pd[s][3][0] = (raw[6] & 0xffffu) & m;
        pd[s][3][1] = (raw[6] >> 16) & m;
    };
    load_d(0);
    convert(std::integral_constant<int, 0>{}, true, 0);
    if (T > 1) load_d(1);

    float acc[27][2];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t][0] = acc[t][1] = 0.f;

    auto frame = [&](auto PH, int f) {
        constexpr int ph = decltype(PH)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // input frame f (this wave's pieces) and the d_conv registers of frame f+1
        __builtin_amdgcn_s_barrier();
        const int cur = P::NBUF == 2 ? (f & 1) : 0;
        convert(std::integral_constant<int, (ph + 1) % 3>{}, f + 1 < T, f + 1);     // d_conv frame f+1 (tap dt = 0) replaces frame f-2
        if (P::NBUF == 2 && f + 1 < T) dma(f + 1, cur ^ 1);
        if (f + 2 < T) load_d(f + 2);
        const char* tile = smem + cur * P::IN_BYTES + cp * 2 * ES;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            uint32_t xr[P::IW];
            const char* rp = tile + (S * row + dy) * P::IW * 96 * ES;
#pragma unroll
            for (int ix = 0; ix < P::IW; ++ix) xr[ix] = *reinterpret_cast<const uint32_t*>(rp + ix * 96 * ES);
            // token pairs of the input row: output tokens (2k, 2k+1) meet inputs (S 2k + dx, S (2k+1) + dx); the 7th token alone
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                uint32_t pi[4][2];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    pi[k][0] = pair_lo(xr[S * 2 * k + dx], xr[S * (2 * k + 1) + dx]);
                    pi[k][1] = pair_hi(xr[S * 2 * k + dx], xr[S * (2 * k + 1) + dx]);
                }
                pi[3][0] = xr[S * 6 + dx] & 0xffffu;
                pi[3][1] = xr[S * 6 + dx] >> 16;
#pragma unroll
                for (int dt = 0; dt < 3; ++dt) {       // input frame f is tap dt of output frame f + 1 - dt
                    const int s = (ph + 4 - dt) % 3, tap = (dt * 3 + dy) * 3 + dx;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        acc[tap][0] = dot2_16(pd[s][k][0], pi[k][0], acc[tap][0]);
                        acc[tap][1] = dot2_16(pd[s][k][1], pi[k][1], acc[tap][1]);
                    }
                }
            }
        }
        if (P::NBUF == 1 && f + 1 < T) {
            __builtin_amdgcn_s_barrier();
            dma(f + 1, 0);
        }
    };
    for (int f0 = 0; f0 < T; f0 += 3) {
        frame(std::integral_constant<int, 0>{}, f0);
        if (f0 + 1 < T) frame(std::integral_constant<int, 1>{}, f0 + 1);
        if (f0 + 2 < T) frame(std::integral_constant<int, 2>{}, f0 + 2);
    }
    // rows added in row order through one [tap][96] slab (aliases the input tiles: every wave is past its last tile read after the barrier)
    float* red = reinterpret_cast<float*>(smem);
    static_assert(27 * 96 * 4 <= P::IN_BYTES, "slab must fit the tile storage");
    for (int r = 0; r < P::ROWS; ++r) {
        __builtin_amdgcn_s_barrier();
        if (row == r && live) {
#pragma unroll
            for (int t = 0; t < 27; ++t) {
                float2* p = reinterpret_cast<float2*>(&red[t * 96 + 2 * cp]);
                float2 v = make_float2(acc[t][0], acc[t][1]);
                if (r > 0) { const float2 o = *p; v.x += o.x; v.y += o.y; }
                *p = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    float* prow = part + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2592;
    for (int i = tid; i < 2592; i += P::NT) {
        const int c = i / 27, t = i - c * 27;          // output index c * 27 + tap (the layout pool_reduce and the conv weight share)
        prow[i] = red[t * 96 + c];
    }
    if constexpr (LNB) {                               // d_gamma | d_beta of the workgroup's tokens: rows added in row order, [192] per workgroup
        float* rl = red;                               // (the [tap][96] slab has been read out)
        for (int r = 0; r < P::ROWS; ++r) {
            __builtin_amdgcn_s_barrier();
            if (row == r && live) {
                float2 a = make_float2(dga0, dga1), b2_ = make_float2(dbe0, dbe1);
                float2* pa = reinterpret_cast<float2*>(&rl[2 * cp]);
                float2* pb = reinterpret_cast<float2*>(&rl[96 + 2 * cp]);
                if (r > 0) { const float2 oa = *pa, ob = *pb; a.x += oa.x; a.y += oa.y; b2_.x += ob.x; b2_.y += ob.y; }
                *pa = a; *pb = b2_;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (tid < 192) part_ln[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 192 + tid] = rl[tid];
    }
}

// weight-gradient partial rows, march form: returns the number of rows written ([2592] each), MVIT_EUNSUPPORTED for shapes it does not take
int mvit_internal_pool_wgrad_march(const void* qkv, int64_t ld, int chan_off, const void* dconv, float* part, int B, int heads, int T, int H,
                                   int W, int stride_hw, int nset, hipStream_t st, const void* xhat, const void* dout, const float* rstd,
                                   const float* gamma, const float* gamma2, float* part_ln) {
    // xhat != nullptr: the fused form (LayerNorm backward in front): `dconv` is then an OUTPUT, part_ln receives [rows][192]
    if (stride_hw != 1 && stride_hw != 2) return MVIT_EUNSUPPORTED;
    if ((int64_t)H * W * ld * 2 >= (1ll << 31)) return MVIT_EUNSUPPORTED;
    const int Ho = (H - 1) / stride_hw + 1, Wo = (W - 1) / stride_hw + 1;
    dim3 grid(((Wo + 6) / 7) * ((Ho + 6) / 7), nset * B * heads);
#define WG_LAUNCH(S, LNB)                                                                                                                 \
    {                                                                                                                                     \
        using P = March<bf16_t, S>;                                                                                                       \
        constexpr int SM = P::NBUF * P::IN_BYTES;                                                                                         \
        static DevFlags attr_tab; DevFlag attr_done = dev_flag(attr_tab);                                                                 \
        if (!attr_done) {                                                                                                                 \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pool_wgrad_march_kernel<S, LNB>), hipFuncAttributeMaxDynamicSharedMemorySize, SM) != hipSuccess) \
                return MVIT_ELAUNCH;                                                                                                      \
            attr_done = true;                                                                                                             \
        }                                                                                                                                 \
        hipLaunchKernelGGL((pool_wgrad_march_kernel<S, LNB>), grid, dim3(P::NT), SM, st, (const bf16_t*)qkv, ld, chan_off, (const bf16_t*)dconv, \
                           part, heads, T, H, W, Ho, Wo, nset == 2 ? B * heads : 0, (const bf16_t*)xhat, (const bf16_t*)dout, rstd, gamma, \
                           gamma2, (bf16_t*)const_cast<void*>(dconv), part_ln);                                                           \
    }
    if (xhat) { if (stride_hw == 1) WG_LAUNCH(1, true) else WG_LAUNCH(2, true) }
    else { if (stride_hw == 1) WG_LAUNCH(1, false) else WG_LAUNCH(2, false) }
#undef WG_LAUNCH
    MVIT_LAUNCH_CHECK();
    return (int)(grid.x * grid.y);
}
// rows mvit_internal_pool_wgrad_march writes for one tensor (workspace sizing)
int64_t mvit_internal_pool_wgrad_march_rows(int B, int heads, int H, int W, int stride_hw) {
    const int Ho = (H - 1) / stride_hw + 1, Wo = (W - 1) / stride_hw + 1;
    return (int64_t)((Wo + 6) / 7) * ((Ho + 6) / 7) * B * heads;
}
