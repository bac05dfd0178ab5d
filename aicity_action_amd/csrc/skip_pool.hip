// Skip path of a stage-transition block (reference: slowfast/models/attention.py:424-432 -- x = proj(x) when the block widens,
// then attention_pool(x, pool_skip) = MaxPool3d k(1,3,3) s(1,2,2) p(0,1,1) -- and its backward).
//
// Unfused, the widened full-resolution fp32 tensor (tokens x Cout: 616 MB at block 1, B = 8 @448) is written by the GEMM, read
// by the max pool, and in training written again by the un-pool, read by the data-gradient GEMM and read by the weight-gradient
// GEMM.  Here it never reaches HBM:
//   forward : one workgroup owns a 4 x 6 patch of POOLED positions of one frame; its GEMM rows are the 9 x 13 input tokens under
//             that patch (gathered rows, halo recomputed: the GEMM is 1 % of the kernel), gathered once and resident in LDS while
//             the workgroup walks the Cout / 96 column tiles; the fp32 products go to an LDS stage and leave as window maxima
//             (+ the arg-max byte per element when training).
//   backward: one workgroup owns an 8 x 16 patch of input tokens; the A operand of the data-gradient GEMM dx = unpool(dy) W is
//             built on chip from the 5 x 9 pooled positions under the patch (one index byte + one gradient per window, ATen's
//             first-maximum rule as recorded by the forward) and is also emitted once, in the 16-bit type, for the
//             weight-gradient GEMM.
// Same MFMA sequence per token as linear_mfma_kernel (96-wide K slabs, 32x32x16, k ascending): results are bit-identical to the
// unfused pair of calls.
#include "common.h"

#define SP_PY 4                       // pooled rows / columns per workgroup
#define SP_PX 6
#define SP_TY (2 * SP_PY + 1)         // 9 x 13 = 117 input tokens (<= 128 GEMM rows)
#define SP_TX (2 * SP_PX + 1)
#define SP_ROWS 128
#define SP_BN 96
#define SP_BK 96
#define SP_ROWB 192                   // bytes per LDS slab row (96 x 16 bit)
#define SP_LD 100                     // backward: fp32 output stage leading dimension (floats), [128][96] = 51200 bytes

// [rows][96] 16-bit slab, 16-byte chunk c of row r at position (c + ((r>>2)&3)) % 12: the 32x32x16 fragment reads are conflict-free
__device__ __forceinline__ int sp_slab_off(int row, int chunk) {
    int p = chunk + ((row >> 2) & 3);
    p = p >= 12 ? p - 12 : p;
    return row * SP_ROWB + p * 16;
}
__device__ __forceinline__ uint4 sp_pack8(const float* p) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    uint4 r;
    r.x = pack_bf16x2(a.x, a.y); r.y = pack_bf16x2(a.z, a.w);
    r.z = pack_bf16x2(b.x, b.y); r.w = pack_bf16x2(b.z, b.w);
    return r;
}
// workgroups with equal blockIdx % 8 (one XCD) take a contiguous range of logical tiles: the n-tiles and neighbouring patches that
// share input tokens meet in one L2
__device__ __forceinline__ int sp_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    const int base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

// 128 x 96 x 96 slab product on the LDS images (4 waves x 32 rows, three 32-column accumulators each)
__device__ __forceinline__ void sp_slab_mfma(const char* fa, const char* fb, const int (&foff)[6], f32x16 (&acc)[3]) {
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(fa + foff[ks]);
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) {
            const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(fb + nb * 32 * SP_ROWB + foff[ks]);
            acc[nb] = mfma16(af, bfr, acc[nb]);
        }
    }
}

// Forward.  The workgroup gathers its 117 token rows ONCE (all of K, rounded to the 16-bit type, resident in LDS) and walks the
// Cout / 96 column tiles: weight slabs stream through one 18 KiB buffer (the next one fetched into registers under the MFMAs),
// the products of a column tile leave 32 columns at a time through an fp32 stage that aliases the weight buffer.
#define SF_LD 36                                   // stage leading dimension (floats): [128 token rows][32 columns]
template <bool IDX, int NK>                        // NK = Cin / 96
__global__ __launch_bounds__(256, NK == 1 ? 3 : (NK == 2 ? 2 : 1)) void proj_maxpool_kernel(
    const float* __restrict__ x, const bf16_t* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y,
    uint32_t* __restrict__ idx, bf16_t* __restrict__ x16, int H, int W, int Ho, int Wo, int Cout, int nty, int ntx) {
    constexpr int Cin = NK * SP_BK;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                                        // NK slabs [128][96]
    char* sB = smem + NK * SP_ROWS * SP_ROWB;               // one weight slab [96][96] | the stage
    float* stage = reinterpret_cast<float*>(sB);

    int tile = sp_xcd_remap(blockIdx.x, gridDim.x);
    const int tx = tile % ntx; tile /= ntx;
    const int ty = tile % nty;
    const int bt = tile / nty;
    const int yo0 = ty * SP_PY, xo0 = tx * SP_PX;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int srow = tid >> 4, schk = tid & 15;
    const bool s_on = schk < 12;
    const int s_lds = sp_slab_off(srow, s_on ? schk : 0);
    const int cch = 8 * (s_on ? schk : 0);

    // gathered GEMM rows: row m = ly * 13 + lx is input token (2 yo0 - 1 + ly, 2 xo0 - 1 + lx) of frame bt; tokens outside the
    // frame (padding) and rows >= 117 read a clamped in-frame token, their products are never used
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int m = srow + 16 * i;
        m = m < SP_TY * SP_TX ? m : 0;
        const int ly = m / SP_TX, lx = m - ly * SP_TX;
        int yi = 2 * yo0 - 1 + ly, xi = 2 * xo0 - 1 + lx;
        yi = yi < 0 ? 0 : (yi >= H ? H - 1 : yi);
        xi = xi < 0 ? 0 : (xi >= W ? W - 1 : xi);
        const uint32_t xo_ = (uint32_t)(((bt * H + yi) * W + xi) * Cin + cch);
        const float* xp = x + xo_;
        uint4 v[NK];
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) v[kt] = sp_pack8(xp + kt * SP_BK);
        if (s_on) {
#pragma unroll
            for (int kt = 0; kt < NK; ++kt) *reinterpret_cast<uint4*>(sA + kt * (SP_ROWS * SP_ROWB) + s_lds + i * 16 * SP_ROWB) = v[kt];
            // training: the rounded rows leave once more as the 16-bit operand of the weight-gradient GEMM -- every token from the patch
            // that holds it in its interior (rows 1..8, columns 1..12 of the 9 x 13 halo grid)
            if (IDX && x16 && ly >= 1 && lx >= 1 && 2 * yo0 - 1 + ly < H && 2 * xo0 - 1 + lx < W && srow + 16 * i < SP_TY * SP_TX) {
#pragma unroll
                for (int kt = 0; kt < NK; ++kt) *reinterpret_cast<uint4*>(x16 + xo_ + kt * SP_BK) = v[kt];
            }
        }
    }
    const bf16_t* w_ptr = w + (int64_t)srow * Cin + cch;
    // Weight slabs go global -> registers -> LDS.  Round 6: TWO slabs of lead for NK >= 2 (two register sets, alternating by slab parity): with
    // one slab of lead the L2 latency of a slab (~1.5 k cycles) stood against the 0.24 us of MFMA of the slab in front of it -- a workgroup at
    // block 14 spent ~59 us on 7.7 us of MFMA.  Same slabs, same order, same MFMA sequence: the results do not change.
    constexpr int LEAD = NK >= 2 ? 2 : 1;
    const int nslab = (Cout / SP_BN) * NK;
    uint4 ra0, ra1, ra2, ra3, ra4, ra5, rb0, rb1, rb2, rb3, rb4, rb5;       // (named registers: arrays handed to helper lambdas ended up in scratch)
    rb0 = rb1 = rb2 = rb3 = rb4 = rb5 = make_uint4(0, 0, 0, 0);
#define SF_WLOAD(R, J) { const int j_ = (J); if (j_ < nslab) { const int nt_ = j_ / NK, kt_ = j_ - nt_ * NK; \
        const bf16_t* wp_ = w_ptr + (int64_t)(nt_ * SP_BN) * Cin + kt_ * SP_BK; \
        R##0 = *reinterpret_cast<const uint4*>(wp_); R##1 = *reinterpret_cast<const uint4*>(wp_ + 16 * Cin); \
        R##2 = *reinterpret_cast<const uint4*>(wp_ + 32 * Cin); R##3 = *reinterpret_cast<const uint4*>(wp_ + 48 * Cin); \
        R##4 = *reinterpret_cast<const uint4*>(wp_ + 64 * Cin); R##5 = *reinterpret_cast<const uint4*>(wp_ + 80 * Cin); } }
#define SF_WSTORE(R) { char* bp_ = sB + s_lds; \
        *reinterpret_cast<uint4*>(bp_) = R##0; *reinterpret_cast<uint4*>(bp_ + 16 * SP_ROWB) = R##1; *reinterpret_cast<uint4*>(bp_ + 32 * SP_ROWB) = R##2; \
        *reinterpret_cast<uint4*>(bp_ + 48 * SP_ROWB) = R##3; *reinterpret_cast<uint4*>(bp_ + 64 * SP_ROWB) = R##4; *reinterpret_cast<uint4*>(bp_ + 80 * SP_ROWB) = R##5; }
    SF_WLOAD(ra, 0)
    if constexpr (LEAD == 2) SF_WLOAD(rb, 1)

    int foff[6];
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
        int p = 2 * ks + h + ((r >> 2) & 3);
        p = p >= 12 ? p - 12 : p;
        foff[ks] = p * 16;
    }
    const char* fa = sA + (32 * wave + r) * SP_ROWB;
    const char* fb = sB + r * SP_ROWB;

    // pooling item of this thread within a 32-column pass: pooled position o = tid / 8 (24 of them), float4 column group tid % 8
    const int po = tid >> 3, pc4 = tid & 7;
    const int poy = po / SP_PX, pox = po - poy * SP_PX;
    const int yo = yo0 + poy, xo = xo0 + pox;
    const bool p_on = po < SP_PY * SP_PX && yo < Ho && xo < Wo;
    uint32_t okmask = 0;                                     // bit ky * 3 + kx: tap inside the frame
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int yi = 2 * yo - 1 + ky, xi = 2 * xo - 1 + kx;
            if (yi >= 0 && yi < H && xi >= 0 && xi < W) okmask |= 1u << (ky * 3 + kx);
        }
    const float* st_rd = stage + ((2 * poy) * SP_TX + 2 * pox) * SF_LD + 4 * pc4;
    const int64_t e_out = (((int64_t)bt * Ho + (p_on ? yo : 0)) * Wo + (p_on ? xo : 0)) * Cout + 4 * pc4;

    const int ntn = Cout / SP_BN;
    for (int nt = 0; nt < ntn; ++nt) {
        f32x16 acc[3];
#pragma unroll
        for (int nb = 0; nb < 3; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
#define SF_SLAB(KT, R) { __syncthreads(); if (s_on) SF_WSTORE(R) __syncthreads(); SF_WLOAD(R, nt * NK + (KT) + LEAD) \
                         sp_slab_mfma(fa + (KT) * (SP_ROWS * SP_ROWB), fb, foff, acc); }
        // (first barrier: sB / the stage are free and, first time round, the A slabs are about to be complete; second: the slab is in place)
        if constexpr (NK == 1) { SF_SLAB(0, ra) }
        else if constexpr (NK == 2) { SF_SLAB(0, ra) SF_SLAB(1, rb) }
        else { SF_SLAB(0, ra) SF_SLAB(1, rb) SF_SLAB(2, ra) SF_SLAB(3, rb) }
#undef SF_SLAB
        // window maxima, 32 columns per pass; the first maximum in ATen's scan order (ky, kx) among the in-frame taps is recorded
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) {
            __syncthreads();
            const float bv = bias ? bias[nt * SP_BN + nb * 32 + r] : 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ml = 32 * wave + (i & 3) + 8 * (i >> 2) + 4 * h;
                stage[ml * SF_LD + r] = acc[nb][i] + bv;
            }
            __syncthreads();
            if (p_on) {
                float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
                uint32_t bx = 0, by = 0, bz = 0, bw = 0;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const uint32_t wpos = ky * 3 + kx;
                        const bool ok = (okmask >> wpos) & 1u;
                        const float4 v = *reinterpret_cast<const float4*>(st_rd + (ky * SP_TX + kx) * SF_LD);
                        if (ok && v.x > m.x) { m.x = v.x; bx = wpos; }
                        if (ok && v.y > m.y) { m.y = v.y; by = wpos; }
                        if (ok && v.z > m.z) { m.z = v.z; bz = wpos; }
                        if (ok && v.w > m.w) { m.w = v.w; bw = wpos; }
                    }
                const int64_t e = e_out + nt * SP_BN + nb * 32;
                *reinterpret_cast<float4*>(y + e) = m;
                if (IDX) idx[e >> 2] = bx | (by << 8) | (bz << 16) | (bw << 24);
            }
        }
    }
#undef SF_WLOAD
#undef SF_WSTORE
}

extern "C" int mvit_proj_maxpool_fwd(const float* x, const void* w, const float* bias, float* y, void* idx, void* x16, int B, int T, int H,
                                     int W, int Cin, int Cout, int act_dtype, void* stream) {
    if (!x || !w || !y || B <= 0 || T <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return MVIT_EINVAL;
    if (act_dtype != MVIT_BF16) return MVIT_EUNSUPPORTED;            // the exact-fp32 path keeps its two separate calls
    if ((Cin != 96 && Cin != 192 && Cin != 384) || Cout % SP_BN) return MVIT_EUNSUPPORTED;
    if ((int64_t)B * T * H * W * Cin >= (1ll << 31)) return MVIT_EUNSUPPORTED;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int nty = (Ho + SP_PY - 1) / SP_PY, ntx = (Wo + SP_PX - 1) / SP_PX;
    const int64_t nwg = (int64_t)B * T * nty * ntx;
    if (nwg >= (1ll << 31)) return MVIT_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
#define SF_GO(IDX_, NK_) { \
        constexpr int smem = NK_ * SP_ROWS * SP_ROWB + SP_ROWS * SF_LD * 4; \
        if (smem > 65536) { \
            static DevFlags attr_done_tab; DevFlag attr_done = dev_flag(attr_done_tab); \
            if (!attr_done) { \
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(&proj_maxpool_kernel<IDX_, NK_>), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) \
                    return MVIT_ELAUNCH; \
                attr_done = true; \
            } \
        } \
        hipLaunchKernelGGL((proj_maxpool_kernel<IDX_, NK_>), dim3((unsigned)nwg), dim3(256), smem, st, x, (const bf16_t*)w, bias, y, (uint32_t*)idx, (bf16_t*)x16, H, W, \
                           Ho, Wo, Cout, nty, ntx); }
#define SF_GO2(NK_) { if (idx) SF_GO(true, NK_) else SF_GO(false, NK_) }
    if (Cin == 96) SF_GO2(1)
    else if (Cin == 192) SF_GO2(2)
    else SF_GO2(4)
#undef SF_GO2
#undef SF_GO
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}

// ----------------------------------------------------------------------------------------------
// Backward: dx[token][Cin] = g[token][:] . Wm, g = un-pooled gradient (token t gets dy of every window whose recorded arg-max
// position is t, windows added in (yo, xo) order like mvit_maxpool_skip_bwd_idx).  One workgroup owns an 8 x 16 patch of INPUT
// tokens of one frame and all NTN = Cin / 96 column tiles of dx.  Per 96-channel slab of Cout the 5 x 9 pooled positions under
// the patch (dy + index bytes, 21 KB) are staged in LDS once -- every pooled element is wanted by up to 9 tokens, gathering them
// from L2 per token ran at the L2's bandwidth -- each thread then builds the g values of its 8 tokens (one patch column) from
// LDS, packs them into the A slab and writes them out once in the 16-bit type (d16) for the weight-gradient GEMM.  The next
// slab's pooled tile is fetched into registers under the MFMAs.  wt = Wm^T [Cin][Cout] in the 16-bit type.
// ----------------------------------------------------------------------------------------------
#define SB_PH 8                        // input-token patch
#define SB_PW 16
#define SB_QH 5                        // pooled positions under it
#define SB_QW 9
#define SB_NQ (SB_QH * SB_QW)          // 45
#define SB_OFF_B (SP_ROWS * SP_ROWB)                   // 24576: weight slab
#define SB_OFF_D (SB_OFF_B + SP_BN * SP_ROWB)          // 43008: pooled dy, fp32 [45][96]
#define SB_OFF_I (SB_OFF_D + SB_NQ * SP_BK * 4)        // 60288: pooled index bytes [45][96]
#define SB_SMEM (SB_OFF_I + SB_NQ * SP_BK)             // 64608 (the epilogue's 51200-byte stage aliases the front)
#define SB_ND ((SB_NQ * SP_BK / 4 + 255) / 256)        // float4 per thread of the dy tile (5)
#define SB_NI ((SB_NQ * SP_BK / 16 + 255) / 256)       // uint4 per thread of the index tile (2)

template <int NTN>
__global__ __launch_bounds__(256, NTN <= 2 ? 2 : 1) void proj_maxpool_bwd_kernel(const uint8_t* __restrict__ idx, const float* __restrict__ dy,
                                                                                 const bf16_t* __restrict__ wt, float* __restrict__ dx,
                                                                                 bf16_t* __restrict__ d16, int H, int W, int Ho, int Wo, int Cin,
                                                                                 int Cout, int npy, int npx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;
    char* sB = smem + SB_OFF_B;
    float* sD = reinterpret_cast<float*>(smem + SB_OFF_D);
    uint8_t* sI = reinterpret_cast<uint8_t*>(smem + SB_OFF_I);
    float* stage = reinterpret_cast<float*>(smem);

    int tile = sp_xcd_remap(blockIdx.x, gridDim.x);
    const int px = tile % npx; tile /= npx;
    const int py = tile % npy;
    const int bt = tile / npy;
    const int y0 = py * SB_PH, x0 = px * SB_PW;
    const int yo0 = y0 >> 1, xo0 = x0 >> 1;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int srow = tid >> 4, schk = tid & 15;           // this thread's GEMM rows srow + 16 i = patch column lx = srow, patch rows i
    const bool s_on = schk < 12;
    const int s_lds = sp_slab_off(srow, s_on ? schk : 0);
    const int cch = 8 * (s_on ? schk : 0);

    // pooled-tile staging map: element e = tid + 256 j of the [45][24] float4 grid (dy) / the [45][6] uint4 grid (index bytes);
    // positions past the frame re-read the last pooled row / column (never selected: the window codes below say so)
    int d_off[SB_ND], i_off[SB_NI];
#pragma unroll
    for (int j = 0; j < SB_ND; ++j) {
        int e = tid + 256 * j;
        e = e < SB_NQ * 24 ? e : SB_NQ * 24 - 1;
        const int q = e / 24, c = e - q * 24;
        int yo = yo0 + q / SB_QW, xo = xo0 + q % SB_QW;
        yo = yo < Ho ? yo : Ho - 1;
        xo = xo < Wo ? xo : Wo - 1;
        d_off[j] = ((bt * Ho + yo) * Wo + xo) * Cout + 4 * c;
    }
#pragma unroll
    for (int j = 0; j < SB_NI; ++j) {
        int e = tid + 256 * j;
        e = e < SB_NQ * 6 ? e : SB_NQ * 6 - 1;
        const int q = e / 6, c = e - q * 6;
        int yo = yo0 + q / SB_QW, xo = xo0 + q % SB_QW;
        yo = yo < Ho ? yo : Ho - 1;
        xo = xo < Wo ? xo : Wo - 1;
        i_off[j] = ((bt * Ho + yo) * Wo + xo) * Cout + 16 * c;
    }
    // (named registers, not arrays: arrays that live across the slab loop's barriers and branches ended up in scratch memory)
    float4 pd0, pd1, pd2, pd3, pd4;
    uint4 pi0, pi1, rb0, rb1, rb2, rb3, rb4, rb5;
    static_assert(SB_ND == 5 && SB_NI == 2, "staging registers are named for a 45-position pooled tile");
#define SB_FETCH(K0) { \
        pd0 = *reinterpret_cast<const float4*>(dy + (uint32_t)(d_off[0] + (K0))); pd1 = *reinterpret_cast<const float4*>(dy + (uint32_t)(d_off[1] + (K0))); \
        pd2 = *reinterpret_cast<const float4*>(dy + (uint32_t)(d_off[2] + (K0))); pd3 = *reinterpret_cast<const float4*>(dy + (uint32_t)(d_off[3] + (K0))); \
        pd4 = *reinterpret_cast<const float4*>(dy + (uint32_t)(d_off[4] + (K0))); \
        pi0 = *reinterpret_cast<const uint4*>(idx + (uint32_t)(i_off[0] + (K0))); pi1 = *reinterpret_cast<const uint4*>(idx + (uint32_t)(i_off[1] + (K0))); }
#define SB_STORE1(J, V) if (tid + 256 * J < SB_NQ * 24) *reinterpret_cast<float4*>(sD + 4 * (tid + 256 * J)) = V;
#define SB_STORE() { SB_STORE1(0, pd0) SB_STORE1(1, pd1) SB_STORE1(2, pd2) SB_STORE1(3, pd3) SB_STORE1(4, pd4) \
        if (tid < SB_NQ * 6) *reinterpret_cast<uint4*>(sI + 16 * tid) = pi0; \
        if (tid + 256 < SB_NQ * 6) *reinterpret_cast<uint4*>(sI + 16 * (tid + 256)) = pi1; }
#define SB_WLOAD(NT, K0) { const bf16_t* wp_ = w_ptr + (int64_t)((NT) * SP_BN) * Cout + (K0); \
        rb0 = *reinterpret_cast<const uint4*>(wp_); rb1 = *reinterpret_cast<const uint4*>(wp_ + (int64_t)16 * Cout); \
        rb2 = *reinterpret_cast<const uint4*>(wp_ + (int64_t)32 * Cout); rb3 = *reinterpret_cast<const uint4*>(wp_ + (int64_t)48 * Cout); \
        rb4 = *reinterpret_cast<const uint4*>(wp_ + (int64_t)64 * Cout); rb5 = *reinterpret_cast<const uint4*>(wp_ + (int64_t)80 * Cout); }
#define SB_WSTORE() { char* bp_ = sB + s_lds; \
        *reinterpret_cast<uint4*>(bp_) = rb0; *reinterpret_cast<uint4*>(bp_ + 16 * SP_ROWB) = rb1; *reinterpret_cast<uint4*>(bp_ + 32 * SP_ROWB) = rb2; \
        *reinterpret_cast<uint4*>(bp_ + 48 * SP_ROWB) = rb3; *reinterpret_cast<uint4*>(bp_ + 64 * SP_ROWB) = rb4; *reinterpret_cast<uint4*>(bp_ + 80 * SP_ROWB) = rb5; }

    // this thread's column: windows xq (tap kx) and, for odd columns, xq + 1 (tap 0)
    const int xi = x0 + srow;
    const int xq = srow >> 1;
    const uint32_t kx_lo = (srow & 1) ? 2u : 1u;
    const bool x2 = (srow & 1) && (xo0 + xq + 1 < Wo);
    const bool col_ok = xi < W;

    const bf16_t* w_ptr = wt + (int64_t)srow * Cout + cch;
    f32x16 acc[NTN][3];
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
        for (int nb = 0; nb < 3; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nt][nb][i] = 0.f;
    int foff[6];
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
        int p = 2 * ks + h + ((r >> 2) & 3);
        p = p >= 12 ? p - 12 : p;
        foff[ks] = p * 16;
    }
    const char* fa = sA + (32 * wave + r) * SP_ROWB;
    const char* fb = sB + r * SP_ROWB;

    // one window's contribution to the 8 channels of this thread: code = tap position the window must have recorded (0xff: none)
    auto add_window = [&](float (&g)[8], int q, uint32_t code) __attribute__((always_inline)) {
        const uint2 ib = *reinterpret_cast<const uint2*>(sI + q * SP_BK + cch);
        const float4 g0 = *reinterpret_cast<const float4*>(sD + q * SP_BK + cch);
        const float4 g1 = *reinterpret_cast<const float4*>(sD + q * SP_BK + cch + 4);
        g[0] += ((ib.x & 0xffu) == code) ? g0.x : 0.f;
        g[1] += (((ib.x >> 8) & 0xffu) == code) ? g0.y : 0.f;
        g[2] += (((ib.x >> 16) & 0xffu) == code) ? g0.z : 0.f;
        g[3] += ((ib.x >> 24) == code) ? g0.w : 0.f;
        g[4] += ((ib.y & 0xffu) == code) ? g1.x : 0.f;
        g[5] += (((ib.y >> 8) & 0xffu) == code) ? g1.y : 0.f;
        g[6] += (((ib.y >> 16) & 0xffu) == code) ? g1.z : 0.f;
        g[7] += ((ib.y >> 24) == code) ? g1.w : 0.f;
    };

    const int nk = Cout / SP_BK;
    SB_FETCH(0)
    for (int kt = 0; kt < nk; ++kt) {
        const int k0 = kt * SP_BK;
        __syncthreads();                      // the previous slab's MFMAs / gathers are done with sA, sB, sD, sI
        SB_STORE()
        SB_WLOAD(0, k0)                       // (idle lanes load too, at a clamped address)
        __syncthreads();
        if (kt + 1 < nk) SB_FETCH(k0 + SP_BK)
        if (s_on) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {            // patch row i: even rows sit under one pooled row (tap ky = 1), odd rows under two
                float g[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = 0.f;
                const int yq = i >> 1;
                const uint32_t ky_lo = (i & 1) ? 2u : 1u;
                const int q00 = yq * SB_QW + xq;
                add_window(g, q00, ky_lo * 3 + kx_lo);
                add_window(g, q00 + 1, x2 ? ky_lo * 3 : 0xffu);
                if (i & 1) {
                    const bool y2 = yo0 + yq + 1 < Ho;
                    add_window(g, q00 + SB_QW, y2 ? kx_lo : 0xffu);
                    add_window(g, q00 + SB_QW + 1, (y2 && x2) ? 0u : 0xffu);
                }
                uint4 v;
                v.x = pack_bf16x2(g[0], g[1]); v.y = pack_bf16x2(g[2], g[3]);
                v.z = pack_bf16x2(g[4], g[5]); v.w = pack_bf16x2(g[6], g[7]);
                *reinterpret_cast<uint4*>(sA + s_lds + i * 16 * SP_ROWB) = v;
                const int yi = y0 + i;
                if (d16 && col_ok && yi < H) *reinterpret_cast<uint4*>(d16 + ((int64_t)(bt * H + yi) * W + xi) * Cout + k0 + cch) = v;
            }
            SB_WSTORE()
        }
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt) {
            __syncthreads();
            if (nt + 1 < NTN) SB_WLOAD(nt + 1, k0)
            sp_slab_mfma(fa, fb, foff, acc[nt]);
            if (nt + 1 < NTN) {
                __syncthreads();
                if (s_on) SB_WSTORE()
            }
        }
    }

    // dx rows: GEMM row m = token (y0 + m / 16, x0 + m % 16)
    const int erow = tid >> 5, ec = tid & 31;       // 32 lanes per row (24 carry a float4), 8 rows per pass
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt) {
        __syncthreads();
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) {
            const int n = nb * 32 + r;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ml = 32 * wave + (i & 3) + 8 * (i >> 2) + 4 * h;
                stage[ml * SP_LD + n] = acc[nt][nb][i];
            }
        }
        __syncthreads();
        if (ec < SP_BN / 4) {
#pragma unroll
            for (int i = 0; i < SP_ROWS / 8; ++i) {
                const int row = erow + 8 * i;
                const int yi = y0 + (row >> 4), xe = x0 + (row & 15);
                if (yi < H && xe < W)
                    *reinterpret_cast<float4*>(dx + ((int64_t)(bt * H + yi) * W + xe) * Cin + nt * SP_BN + 4 * ec) =
                        *reinterpret_cast<const float4*>(stage + row * SP_LD + 4 * ec);
            }
        }
    }
}

#undef SB_FETCH
#undef SB_STORE1
#undef SB_STORE
#undef SB_WLOAD
#undef SB_WSTORE

extern "C" int mvit_proj_maxpool_bwd(const void* idx, const float* dy, const void* wt, float* dx, void* d16, int B, int T, int H,
                                     int W, int Cin, int Cout, int act_dtype, void* stream) {
    if (!idx || !dy || !wt || !dx || B <= 0 || T <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return MVIT_EINVAL;
    if (act_dtype != MVIT_BF16) return MVIT_EUNSUPPORTED;
    if (Cout % SP_BK || (Cin != 96 && Cin != 192 && Cin != 384)) return MVIT_EUNSUPPORTED;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    if ((int64_t)B * T * H * W >= (1ll << 31) || (int64_t)B * T * Ho * Wo * Cout >= (1ll << 30)) return MVIT_EUNSUPPORTED;   // 32-bit offsets
    const int npy = (H + SB_PH - 1) / SB_PH, npx = (W + SB_PW - 1) / SB_PW;
    const int64_t nwg = (int64_t)B * T * npy * npx;
    if (nwg >= (1ll << 31)) return MVIT_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
#define SP_BWD(NTN_) hipLaunchKernelGGL(proj_maxpool_bwd_kernel<NTN_>, dim3((unsigned)nwg), dim3(256), SB_SMEM, st, (const uint8_t*)idx, dy, \
                                        (const bf16_t*)wt, dx, (bf16_t*)d16, H, W, Ho, Wo, Cin, Cout, npy, npx)
    if (Cin == 96) SP_BWD(1);
    else if (Cin == 192) SP_BWD(2);
    else SP_BWD(4);
#undef SP_BWD
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}
