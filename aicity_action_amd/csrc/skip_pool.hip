// Skip path of a stage-transition block (reference: slowfast/models/attention.py:424-432 -- x = proj(x) when the block widens,
// then attention_pool(x, pool_skip) = MaxPool3d k(1,3,3) s(1,2,2) p(0,1,1) -- and its backward).
//
// Unfused, the widened full-resolution fp32 tensor (tokens x Cout: 616 MB at block 1, B = 8 @448) is written by the GEMM, read
// by the max pool, and in training written again by the un-pool, read by the data-gradient GEMM and read by the weight-gradient
// GEMM.  Here it never reaches HBM:
//   forward : one workgroup owns a 4 x 6 patch of POOLED positions of one frame and 96 output channels; its GEMM rows are the
//             9 x 13 input tokens under that patch (gathered rows, halo recomputed: the GEMM is 1 % of the kernel), the fp32
//             products go to an LDS stage and leave as window maxima (+ the arg-max byte per element when training).
//   backward: the A operand of the data-gradient GEMM dx = unpool(dy) W is built on the fly -- each input token gathers the <= 4
//             windows it belongs to (one index byte + one gradient each, ATen's first-maximum rule as recorded by the forward) --
//             and is also emitted once, in the 16-bit type, for the weight-gradient GEMM.
// Same MFMA sequence per token as linear_mfma_kernel (96-wide K slabs, 32x32x16, k ascending): results are bit-identical to the
// unfused pair of calls.
#include "common.h"

#ifndef SP_ABL
#define SP_ABL 0      // timing ablations (results invalid): 1 no window scan, 2 no global loads, 4 no MFMA, 8 no stage writes, 16 no output stores
#endif
#define SP_PY 4                       // pooled rows / columns per workgroup
#define SP_PX 6
#define SP_TY (2 * SP_PY + 1)         // 9 x 13 = 117 input tokens (<= 128 GEMM rows)
#define SP_TX (2 * SP_PX + 1)
#define SP_ROWS 128
#define SP_BN 96
#define SP_BK 96
#define SP_ROWB 192                   // bytes per LDS slab row (96 x 16 bit)
#define SP_LD 100                     // fp32 stage leading dimension (floats)
#define SP_SMEM (SP_ROWS * SP_LD * 4) // 51200 >= the two slabs (43008)

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

template <bool IDX>
__global__ __launch_bounds__(256, 3) void proj_maxpool_kernel(const float* __restrict__ x, const bf16_t* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ y,
                                                              uint32_t* __restrict__ idx, int H, int W, int Ho, int Wo, int Cin,
                                                              int Cout, int nty, int ntx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;
    char* sB = smem + SP_ROWS * SP_ROWB;
    float* stage = reinterpret_cast<float*>(smem);

    const int ntn = Cout / SP_BN;
    int tile = sp_xcd_remap(blockIdx.x, gridDim.x);
    const int tn = tile % ntn; tile /= ntn;
    const int tx = tile % ntx; tile /= ntx;
    const int ty = tile % nty;
    const int bt = tile / nty;
    const int n0 = tn * SP_BN, yo0 = ty * SP_PY, xo0 = tx * SP_PX;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int srow = tid >> 4, schk = tid & 15;
    const bool s_on = schk < 12;
    const int s_lds = sp_slab_off(srow, s_on ? schk : 0);

    // gathered GEMM rows: row m = ly * 13 + lx is input token (2 yo0 - 1 + ly, 2 xo0 - 1 + lx) of frame bt; tokens outside the
    // frame (padding) and rows >= 117 read a clamped in-frame token, their products are never used
    int a_off[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int m = srow + 16 * i;
        m = m < SP_TY * SP_TX ? m : 0;
        const int ly = m / SP_TX, lx = m - ly * SP_TX;
        int yi = 2 * yo0 - 1 + ly, xi = 2 * xo0 - 1 + lx;
        yi = yi < 0 ? 0 : (yi >= H ? H - 1 : yi);
        xi = xi < 0 ? 0 : (xi >= W ? W - 1 : xi);
        a_off[i] = ((bt * H + yi) * W + xi) * Cin + 8 * (s_on ? schk : 0);
    }
    const bf16_t* w_ptr = w + (int64_t)(n0 + srow) * Cin + 8 * (s_on ? schk : 0);

    f32x16 acc[3];
#pragma unroll
    for (int nb = 0; nb < 3; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
    int foff[6];
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
        int p = 2 * ks + h + ((r >> 2) & 3);
        p = p >= 12 ? p - 12 : p;
        foff[ks] = p * 16;
    }
    const char* fa = sA + (32 * wave + r) * SP_ROWB;
    const char* fb = sB + r * SP_ROWB;

    // A rows of the next slab are fetched under the MFMAs of this one; the weight slab (L2-resident) is loaded and stored in place
    uint4 ra[8];
    const int nk = Cin / SP_BK;
    if (s_on) {
#pragma unroll
        for (int i = 0; i < 8; ++i) ra[i] = (SP_ABL & 2) ? make_uint4(a_off[i], 0, i, 1) : sp_pack8(x + a_off[i]);
    }
    for (int kt = 0; kt < nk; ++kt) {
        const int k0 = kt * SP_BK;
        __syncthreads();
        if (s_on) {
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4*>(sA + s_lds + i * 16 * SP_ROWB) = ra[i];
#pragma unroll
            for (int i = 0; i < 6; ++i)
                *reinterpret_cast<uint4*>(sB + s_lds + i * 16 * SP_ROWB) =
                    (SP_ABL & 2) ? make_uint4(k0, i, 3, 1) : *reinterpret_cast<const uint4*>(w_ptr + (int64_t)i * 16 * Cin + k0);
        }
        __syncthreads();
        if (kt + 1 < nk && s_on) {
#pragma unroll
            for (int i = 0; i < 8; ++i) ra[i] = (SP_ABL & 2) ? make_uint4(a_off[i], k0, i, 1) : sp_pack8(x + a_off[i] + k0 + SP_BK);
        }
        if (!(SP_ABL & 4)) sp_slab_mfma(fa, fb, foff, acc);
    }

    // products (+ bias) -> fp32 stage [token row][96]
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) {
        const int n = nb * 32 + r;
        const float bv = bias ? bias[n0 + n] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ml = 32 * wave + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (!(SP_ABL & 8)) stage[ml * SP_LD + n] = acc[nb][i] + bv;
        }
    }
    __syncthreads();
    // window maxima: 24 pooled positions x 24 float4 column groups; the first maximum in ATen's scan order (ky, kx) among the
    // in-frame taps is the one recorded
    for (int item = tid; item < SP_PY * SP_PX * (SP_BN / 4); item += 256) {
        const int o = item / (SP_BN / 4), c4 = item - o * (SP_BN / 4);
        const int oy = o / SP_PX, ox = o - oy * SP_PX;
        const int yo = yo0 + oy, xo = xo0 + ox;
        if (yo >= Ho || xo >= Wo) continue;
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        uint32_t bx = 0, by = 0, bz = 0, bw = 0;
        if (SP_ABL & 1) m = *reinterpret_cast<const float4*>(stage + ((2 * oy + 1) * SP_TX + 2 * ox + 1) * SP_LD + 4 * c4);
#pragma unroll
        for (int ky = 0; ky < ((SP_ABL & 1) ? 0 : 3); ++ky) {
            const int yi = 2 * yo - 1 + ky;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xi = 2 * xo - 1 + kx;
                const bool ok = yi >= 0 && yi < H && xi >= 0 && xi < W;
                const float4 v = *reinterpret_cast<const float4*>(stage + ((2 * oy + ky) * SP_TX + 2 * ox + kx) * SP_LD + 4 * c4);
                const uint32_t wpos = ky * 3 + kx;
                if (ok && v.x > m.x) { m.x = v.x; bx = wpos; }
                if (ok && v.y > m.y) { m.y = v.y; by = wpos; }
                if (ok && v.z > m.z) { m.z = v.z; bz = wpos; }
                if (ok && v.w > m.w) { m.w = v.w; bw = wpos; }
            }
        }
        const int64_t e = (((int64_t)bt * Ho + yo) * Wo + xo) * Cout + n0 + 4 * c4;
        if ((SP_ABL & 16) && m.x != 12345.f) continue;
        *reinterpret_cast<float4*>(y + e) = m;
        if (IDX) idx[e >> 2] = bx | (by << 8) | (bz << 16) | (bw << 24);
    }
}

extern "C" int mvit_proj_maxpool_fwd(const float* x, const void* w, const float* bias, float* y, void* idx, int B, int T, int H,
                                     int W, int Cin, int Cout, int act_dtype, void* stream) {
    if (!x || !w || !y || B <= 0 || T <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return MVIT_EINVAL;
    if (act_dtype != MVIT_BF16) return MVIT_EUNSUPPORTED;            // the exact-fp32 path keeps its two separate calls
    if (Cin % SP_BK || Cout % SP_BN) return MVIT_EUNSUPPORTED;
    if ((int64_t)B * T * H * W * Cin >= (1ll << 31)) return MVIT_EUNSUPPORTED;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int nty = (Ho + SP_PY - 1) / SP_PY, ntx = (Wo + SP_PX - 1) / SP_PX;
    const int64_t nwg = (int64_t)B * T * nty * ntx * (Cout / SP_BN);
    if (nwg >= (1ll << 31)) return MVIT_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
    if (idx)
        hipLaunchKernelGGL(proj_maxpool_kernel<true>, dim3((unsigned)nwg), dim3(256), SP_SMEM, st, x, (const bf16_t*)w, bias, y,
                           (uint32_t*)idx, H, W, Ho, Wo, Cin, Cout, nty, ntx);
    else
        hipLaunchKernelGGL(proj_maxpool_kernel<false>, dim3((unsigned)nwg), dim3(256), SP_SMEM, st, x, (const bf16_t*)w, bias, y,
                           (uint32_t*)nullptr, H, W, Ho, Wo, Cin, Cout, nty, ntx);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}

// ----------------------------------------------------------------------------------------------
// Backward: dx[token][Cin] = g[token][:] . Wm, g = un-pooled gradient (token t gets dy of every window whose recorded arg-max
// position is t, windows added in (yo, xo) order like mvit_maxpool_skip_bwd_idx).  g is the A operand, built in registers from
// one index byte + one gradient per (window, channel), once per 128-token tile: the workgroup carries the accumulators of all
// NTN = Cin / 96 column tiles; it also writes g out in the 16-bit type (d16) for the weight-gradient GEMM.
// wt = Wm^T [Cin][Cout] in the 16-bit type.
// ----------------------------------------------------------------------------------------------
template <int NTN>
__global__ __launch_bounds__(256, NTN == 1 ? 3 : (NTN == 2 ? 2 : 1)) void proj_maxpool_bwd_kernel(const uint8_t* __restrict__ idx, const float* __restrict__ dy,
                                                                  const bf16_t* __restrict__ wt, float* __restrict__ dx,
                                                                  bf16_t* __restrict__ d16, int64_t tokens, int H, int W, int Ho,
                                                                  int Wo, int Cin, int Cout) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;
    char* sB = smem + SP_ROWS * SP_ROWB;          // NTN weight slabs back to back
    float* stage = reinterpret_cast<float*>(smem);

    const int tile = sp_xcd_remap(blockIdx.x, gridDim.x);
    const int64_t m0 = (int64_t)tile * SP_ROWS;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int srow = tid >> 4, schk = tid & 15;
    const bool s_on = schk < 12;
    const int s_lds = sp_slab_off(srow, s_on ? schk : 0);
    const int cch = 8 * (s_on ? schk : 0);

    // per gathered row: element offset of its first window (yo_lo, xo_lo) and the tap codes of its <= 4 windows (0xff = no window)
    int e0[8];
    uint32_t me4[8];          // bytes: (lo,lo) (lo,hi) (hi,lo) (hi,hi)
    bool row_ok[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int64_t t64 = m0 + srow + 16 * i;
        row_ok[i] = t64 < tokens;
        const uint32_t t = (uint32_t)(row_ok[i] ? t64 : tokens - 1);          // tokens < 2^31 (host check)
        const uint32_t q = t / (uint32_t)W;
        const int xi = (int)(t - q * (uint32_t)W);
        const int bt = (int)(q / (uint32_t)H);
        const int yi = (int)(q - (uint32_t)bt * (uint32_t)H);
        const int yo = yi >> 1, xo = xi >> 1;
        const uint32_t ky = (yi & 1) ? 2u : 1u, kx = (xi & 1) ? 2u : 1u;
        const bool y2 = (yi & 1) && yo + 1 < Ho, x2 = (xi & 1) && xo + 1 < Wo;
        e0[i] = ((bt * Ho + yo) * Wo + xo) * Cout + cch;
        const uint32_t c00 = ky * 3 + kx, c01 = x2 ? ky * 3 : 0xffu, c10 = y2 ? kx : 0xffu, c11 = (y2 && x2) ? 0u : 0xffu;
        me4[i] = c00 | (c01 << 8) | (c10 << 16) | (c11 << 24);
    }
    const bf16_t* w_ptr = wt + (int64_t)srow * Cout + cch;
    const int dyo = Wo * Cout;
    const bool emit = d16 != nullptr;

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

    // The gathered rows go straight to the LDS slab (no register staging, no prefetch under the MFMAs: the product is 1 % of this
    // kernel, three workgroups per CU overlap each other's gathers), two rows = 24 loads in flight per thread.
    const int nk = Cout / SP_BK;
    for (int kt = 0; kt < nk; ++kt) {
        const int k0 = kt * SP_BK;
        __syncthreads();
        if (s_on) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float g[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = 0.f;
                int e0i = e0[i];
                uint32_t me = me4[i];
                asm volatile("" : "+v"(e0i), "+v"(me));     // the 32 (row, window) offsets / codes are rebuilt per slab, not kept in 100 registers
                // windows that do not exist re-read the first one (always in range) under a code no byte can match
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t code = (me >> (8 * j)) & 0xffu;
                    // unsigned 32-bit offsets from the (uniform) base pointers: scalar-base addressing, no 64-bit lane addresses to keep
                    const uint32_t off = (uint32_t)(e0i + k0 + (code == 0xffu ? 0 : ((j & 2) ? dyo : 0) + ((j & 1) ? Cout : 0)));
                    const uint2 ib = *reinterpret_cast<const uint2*>(idx + off);
                    const char* gp = reinterpret_cast<const char*>(dy) + 4u * off;
                    const float4 g0 = *reinterpret_cast<const float4*>(gp);
                    const float4 g1 = *reinterpret_cast<const float4*>(gp + 16);
                    g[0] += ((ib.x & 0xffu) == code) ? g0.x : 0.f;
                    g[1] += (((ib.x >> 8) & 0xffu) == code) ? g0.y : 0.f;
                    g[2] += (((ib.x >> 16) & 0xffu) == code) ? g0.z : 0.f;
                    g[3] += ((ib.x >> 24) == code) ? g0.w : 0.f;
                    g[4] += ((ib.y & 0xffu) == code) ? g1.x : 0.f;
                    g[5] += (((ib.y >> 8) & 0xffu) == code) ? g1.y : 0.f;
                    g[6] += (((ib.y >> 16) & 0xffu) == code) ? g1.z : 0.f;
                    g[7] += ((ib.y >> 24) == code) ? g1.w : 0.f;
                }
                uint4 v;
                v.x = pack_bf16x2(g[0], g[1]); v.y = pack_bf16x2(g[2], g[3]);
                v.z = pack_bf16x2(g[4], g[5]); v.w = pack_bf16x2(g[6], g[7]);
                *reinterpret_cast<uint4*>(sA + s_lds + i * 16 * SP_ROWB) = v;
                if (emit && row_ok[i]) *reinterpret_cast<uint4*>(d16 + (m0 + srow + 16 * i) * Cout + k0 + cch) = v;
                if (i & 1) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) {
                uint4 rb[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) rb[i] = *reinterpret_cast<const uint4*>(w_ptr + (int64_t)(nt * SP_BN + i * 16) * Cout + k0);
#pragma unroll
                for (int i = 0; i < 6; ++i) *reinterpret_cast<uint4*>(sB + nt * (SP_BN * SP_ROWB) + s_lds + i * 16 * SP_ROWB) = rb[i];
            }
        }
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt) sp_slab_mfma(fa, fb + nt * (SP_BN * SP_ROWB), foff, acc[nt]);
    }

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
                const int64_t m = m0 + row;
                if (m < tokens)
                    *reinterpret_cast<float4*>(dx + m * Cin + nt * SP_BN + 4 * ec) = *reinterpret_cast<const float4*>(stage + row * SP_LD + 4 * ec);
            }
        }
    }
}

extern "C" int mvit_proj_maxpool_bwd(const void* idx, const float* dy, const void* wt, float* dx, void* d16, int B, int T, int H,
                                     int W, int Cin, int Cout, int act_dtype, void* stream) {
    if (!idx || !dy || !wt || !dx || B <= 0 || T <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return MVIT_EINVAL;
    if (act_dtype != MVIT_BF16) return MVIT_EUNSUPPORTED;
    if (Cin % SP_BN || Cout % SP_BK || (Cin != 96 && Cin != 192 && Cin != 384)) return MVIT_EUNSUPPORTED;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int64_t tokens = (int64_t)B * T * H * W;
    if (tokens >= (1ll << 31)) return MVIT_EUNSUPPORTED;
    if ((int64_t)B * T * Ho * Wo * Cout >= (1ll << 30)) return MVIT_EUNSUPPORTED;      // 32-bit byte offsets into dy
    const int64_t nwg = (tokens + SP_ROWS - 1) / SP_ROWS;
    hipStream_t st = as_stream(stream);
#define SP_BWD(NTN_) { \
        constexpr int smem = SP_ROWS * SP_ROWB + NTN_ * SP_BN * SP_ROWB > SP_SMEM ? SP_ROWS * SP_ROWB + NTN_ * SP_BN * SP_ROWB : SP_SMEM; \
        if (smem > 65536) { \
            static bool attr_done = false; \
            if (!attr_done) { \
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(&proj_maxpool_bwd_kernel<NTN_>), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return MVIT_ELAUNCH; \
                attr_done = true; \
            } \
        } \
        hipLaunchKernelGGL(proj_maxpool_bwd_kernel<NTN_>, dim3((unsigned)nwg), dim3(256), smem, st, (const uint8_t*)idx, dy, (const bf16_t*)wt, dx, \
                           (bf16_t*)d16, tokens, H, W, Ho, Wo, Cin, Cout); }
    if (Cin == 96) SP_BWD(1)
    else if (Cin == 192) SP_BWD(2)
    else SP_BWD(4)
#undef SP_BWD
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}
