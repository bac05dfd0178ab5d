// Pooling conv (depthwise 3x3x3, stride (1,S,S), zero pad 1, one 96-channel kernel shared by all heads) + LayerNorm(96),
// spatial strides 1 and 2 (98 % of the pooled tokens): the "march" kernel.
//   reference: attention_pool, slowfast/models/attention.py:12-83 (conv variant) with norm_{q,k,v} of :185,199,213.
//
// HBM-bound op (192 B in + 192 B out per token at stride 1) that the first version ran at 18 % of the HBM rate because every
// frame paid an exposed global-load round trip, four workgroup barriers and an fp32 LDS round trip for the LayerNorm.
// Structure here:
//   * workgroup = 7 waves = a 7 x 7 output tile of one (batch, head); wave = one output row, lane = one channel PAIR (48 of
//     the 64 lanes carry data; every token grid of the model -- 112, 56, 28, 14, 7 -- is a multiple of 7, so no tile is partial);
//   * the workgroup marches over the T input frames; a frame's halo tile (IH x IW tokens x 96 channels) arrives in LDS by
//     global_load_lds (inline asm: SGPR frame base + 32-bit lane offset), double-buffered so the next frame is in flight under
//     the current frame's arithmetic; halo cells outside the image are zeroed once and never written again;
//   * three rolling accumulator sets (output frames f-1, f, f+1): each LDS value is read once per dy and used for 9 taps,
//     arithmetic in plain v_fma_f32 (measured: the packed v_pk_fma_f32 form of the same loop ran slower);
//   * the LayerNorm of a finished frame runs IN REGISTERS: the 96 channels of a token are the 48 lanes of one wave, so mean and
//     variance are two 64-lane butterflies (DPP quad_perm / row_half_mirror / row_mirror, ds_swizzle xor 16, v_permlane32_swap)
//     -- no LDS stage, no barrier, all lanes busy; one s_barrier per frame in total.
// MODE 0: forward.  MODE 1: training forward, also writes xhat = (conv - mean) * rstd and rstd (what the LayerNorm backward
// needs).  MODE 2 ("plain", stride 1): the bare convolution with mirrored taps written token-major into a channel slice of a
// [B][tokens][out_ld] buffer = the DATA gradient of the stride-1 pooling conv (attention.py:56 backward).
#include <type_traits>

#include "common.h"

typedef __attribute__((ext_vector_type(2))) float f32x2;

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

// sums over the 64 lanes of a wave of N independent values, every lane gets every total.  Level by level over all N values:
// the N butterflies interleave, so no DPP operand is read in the two wait states behind the VALU write that produced it (called
// per value, every v_add_f32_dpp carried an s_nop 1).
template <int N>
__device__ __forceinline__ void wave_sum_n(float (&v)[N]) {
#define DPP_LEVEL(CTRL)                                                                                           \
    _Pragma("unroll") for (int i = 0; i < N; ++i)                                                                 \
        v[i] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[i]), CTRL, 0xF, 0xF, false));
    DPP_LEVEL(0xB1)      // quad_perm [1,0,3,2]
    DPP_LEVEL(0x4E)      // quad_perm [2,3,0,1]
    DPP_LEVEL(0x141)     // row_half_mirror
    DPP_LEVEL(0x140)     // row_mirror
#undef DPP_LEVEL
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v[i]), 0x401F));      // lane ^ 16
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i]), false, false);
        v[i] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
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
                                                         int out_chan_off, int out_heads) {
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
    const int b = bh / heads, g = bh - b * heads;
    const int64_t Nin = (int64_t)T * H * W;
    const char* base = reinterpret_cast<const char*>(in + (int64_t)b * Nin * ld + chan_off + g * 96);
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
    auto finalize = [&](int fo) {        // arithmetic only
        pend_fo = fo;
        if (!row_ok) return;             // wave-uniform
        if constexpr (MODE == 2) {
#pragma unroll
            for (int x = 0; x < P::XO; ++x) pend_o[x] = pack(acc[0][x].x, acc[0][x].y);
        } else {
            float mean[P::XO], rstd[P::XO];
#pragma unroll
            for (int x = 0; x < P::XO; ++x) mean[x] = live ? acc[0][x].x + acc[0][x].y : 0.f;
            wave_sum_n<P::XO>(mean);
#pragma unroll
            for (int x = 0; x < P::XO; ++x) {
                mean[x] *= (1.0f / 96.0f);
                const float d0 = acc[0][x].x - mean[x], d1 = acc[0][x].y - mean[x];
                acc[0][x].x = d0;
                acc[0][x].y = d1;
                rstd[x] = live ? d0 * d0 + d1 * d1 : 0.f;
            }
            wave_sum_n<P::XO>(rstd);
#pragma unroll
            for (int x = 0; x < P::XO; ++x) {
                rstd[x] = __builtin_amdgcn_rsqf(rstd[x] * (1.0f / 96.0f) + eps);       // v_rsq_f32 (1 ulp): one instruction, not the
                const float h0 = acc[0][x].x * rstd[x], h1 = acc[0][x].y * rstd[x];     // 25-instruction IEEE sqrt + divide, per lane
                if constexpr (MODE == 1) pend_h[x] = pack(h0, h1);
                pend_o[x] = pack(fmaf(h0, g2.x, b2.x), fmaf(h1, g2.y, b2.y));
            }
            if constexpr (MODE == 1) {
                pend_r = rstd[0];
#pragma unroll
                for (int x = 1; x < P::XO; ++x) pend_r = lane == x ? rstd[x] : pend_r;
            }
        }
    };
    auto flush = [&]() {                 // stores of the pending frame
        if (pend_fo < 0 || !row_ok) return;
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
    for (int f = 0; f < T; ++f) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // frame f has landed (this wave's pieces) ...
        __builtin_amdgcn_s_barrier();                              // ... and everyone's; all waves are past frame f-1's reads
        const int cur = P::NBUF == 2 ? (f & 1) : 0;
        if (P::NBUF == 2 && f + 1 < T) dma(f + 1, cur ^ 1);
        flush();                           // output frame f-2
        const char* tile = smem + cur * P::IN_BYTES + cp * 2 * ES;
        // keep the 27 weight reads inside the loop (hoisted they would pin 54 registers) -- as LDS reads (see lds_opaque)
        const lds_cptr_t wm = lds_opaque(reinterpret_cast<const char*>(wl) + cp * 8);
#pragma unroll 1
        for (int dy = 0; dy < 3; ++dy) {
            f32x2 xin[P::IW];
            const char* rp = tile + (S * row + dy) * P::IW * 96 * ES;
#pragma unroll
            for (int ix = 0; ix < P::IW; ++ix) xin[ix] = lds_pair<TA>(rp + ix * 96 * ES);
#pragma unroll
            for (int dt = 0; dt < 3; ++dt)      // input frame f is tap dt of output frame f + 1 - dt -> accumulator set 2 - dt
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const f32x2 wv = lds_ld<f32x2>(wm, ((dt * 3 + dy) * 3 + dx) * 384);
#pragma unroll
                    for (int x = 0; x < P::XO; ++x) {      // plain v_fma_f32 x 2: v_pk_fma_f32 issues at well under half their rate on gfx950
                        acc[2 - dt][x].x = fmaf(wv.x, xin[S * x + dx].x, acc[2 - dt][x].x);
                        acc[2 - dt][x].y = fmaf(wv.y, xin[S * x + dx].y, acc[2 - dt][x].y);
                    }
                }
        }
        if (P::NBUF == 1 && f + 1 < T) {   // single buffer: the next frame may only be requested once every wave has read this one
            __builtin_amdgcn_s_barrier();
            dma(f + 1, 0);
        }
        if (f >= 1) finalize(f - 1);       // set 0 = output frame f-1 is complete
#pragma unroll
        for (int x = 0; x < P::XO; ++x) {
            acc[0][x] = acc[1][x];
            acc[1][x] = acc[2][x];
            acc[2][x] = f32x2{0.f, 0.f};
        }
    }
    flush();                               // output frame T-2
    finalize(T - 1);
    flush();
}

template <typename TA, int S, int MODE>
static int launch_march(const void* in, int64_t ld, int chan_off, const float* w, const float* gamma, const float* beta, void* out,
                        void* xhat, float* rstd, int B, int heads, int T, int H, int W, int Ho, int Wo, float eps, int64_t out_ld,
                        int out_chan_off, int out_heads, hipStream_t st) {
    using P = March<TA, S>;
    dim3 grid(((Wo + P::XO - 1) / P::XO) * ((Ho + P::ROWS - 1) / P::ROWS), B * heads);
    static DevFlags attr_done_tab; DevFlag attr_done = dev_flag(attr_done_tab);
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pool_march_kernel<TA, S, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                P::SMEM) != hipSuccess)
            return MVIT_ELAUNCH;
        attr_done = true;
    }
    hipLaunchKernelGGL((pool_march_kernel<TA, S, MODE>), grid, dim3(P::NT), P::SMEM, st, (const TA*)in, ld, chan_off, w, gamma, beta,
                       (TA*)out, (TA*)xhat, rstd, heads, T, H, W, Ho, Wo, eps, out_ld, out_chan_off, out_heads);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}

// internal entry points (pool.hip / pool_bwd.hip): forward (+ saved statistics) for strides 1 and 2 ...
int mvit_internal_pool_march_fwd(const void* qkv, int64_t ld, int chan_off, const float* w, const float* gamma, const float* beta, void* out,
                                 void* xhat, float* rstd, int B, int heads, int T, int H, int W, int stride_hw, float eps, int act_dtype,
                                 hipStream_t st) {
    const int Ho = (H - 1) / stride_hw + 1, Wo = (W - 1) / stride_hw + 1;
    // frame / tile offsets are 32-bit byte offsets from a per-(batch, head) 64-bit base
    if ((int64_t)H * W * ld * 4 >= (1ll << 31)) return MVIT_EUNSUPPORTED;
#define MARCH(TA, S)                                                                                                                   \
    (xhat ? launch_march<TA, S, 1>(qkv, ld, chan_off, w, gamma, beta, out, xhat, rstd, B, heads, T, H, W, Ho, Wo, eps, 0, 0, 1, st)    \
          : launch_march<TA, S, 0>(qkv, ld, chan_off, w, gamma, beta, out, nullptr, nullptr, B, heads, T, H, W, Ho, Wo, eps, 0, 0, 1, st))
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
