// Cube embedding: Conv3d(3->96, k(3,7,7), s(2,4,4), p(1,3,3)) + bias + separable position embedding,
// written token-major [B][T'*H'*W'][96] fp32 (the residual stream).
//
// v1 (exact fp32 VALU): an 8x8 output-token tile per 256-thread workgroup; the (ci,dt) input plane
// patch (35x35) and its 49x96 weight slab are staged in LDS; thread = (token, 24-channel group), so
// weight reads are wave-uniform broadcasts.
#include "common.h"

#define ST_TY 8
#define ST_TX 8
#define ST_PH (4 * ST_TY + 3)  // 35
#define ST_PW (4 * ST_TX + 3)  // 35

__global__ __launch_bounds__(256) void stem_f32_kernel(const float* __restrict__ clip, const float* __restrict__ w,
                                                       const float* __restrict__ bias, const float* __restrict__ pos_s,
                                                       const float* __restrict__ pos_t, float* __restrict__ x, int T,
                                                       int S, int To, int So) {
    __shared__ float patch[ST_PH * ST_PW];
    __shared__ __attribute__((aligned(16))) float wsl[49 * 96];
    const int tiles_x = (So + ST_TX - 1) / ST_TX;
    const int tx0 = (blockIdx.x % tiles_x) * ST_TX;
    const int ty0 = (blockIdx.x / tiles_x) * ST_TY;
    const int to = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x;
    const int tok = tid & 63, cg = tid >> 6;
    const int ly = tok >> 3, lx = tok & 7;
    float acc[24];
#pragma unroll
    for (int e = 0; e < 24; ++e) acc[e] = 0.f;
    for (int ci = 0; ci < 3; ++ci)
        for (int dt = 0; dt < 3; ++dt) {
            const int ti = 2 * to + dt - 1;
            __syncthreads();
            const bool t_ok = ti >= 0 && ti < T;
            for (int i = tid; i < ST_PH * ST_PW; i += 256) {
                const int py = i / ST_PW, px = i - py * ST_PW;
                const int yi = 4 * ty0 + py - 3, xi = 4 * tx0 + px - 3;
                float v = 0.f;
                if (t_ok && yi >= 0 && yi < S && xi >= 0 && xi < S)
                    v = clip[((((int64_t)b * 3 + ci) * T + ti) * S + yi) * S + xi];
                patch[i] = v;
            }
            for (int i = tid; i < 49 * 96; i += 256) {
                const int tap = i / 96, c = i - tap * 96;
                wsl[i] = w[((c * 3 + ci) * 3 + dt) * 49 + tap];
            }
            __syncthreads();
            if (!t_ok) continue;
#pragma unroll 1
            for (int dy = 0; dy < 7; ++dy)
#pragma unroll
                for (int dx = 0; dx < 7; ++dx) {
                    const float v = patch[(4 * ly + dy) * ST_PW + 4 * lx + dx];
                    const float* wt = wsl + (dy * 7 + dx) * 96 + cg * 24;
#pragma unroll
                    for (int e = 0; e < 24; e += 4) {
                        const float4 ww = *reinterpret_cast<const float4*>(wt + e);
                        acc[e] = fmaf(v, ww.x, acc[e]); acc[e + 1] = fmaf(v, ww.y, acc[e + 1]);
                        acc[e + 2] = fmaf(v, ww.z, acc[e + 2]); acc[e + 3] = fmaf(v, ww.w, acc[e + 3]);
                    }
                }
        }
    const int yo = ty0 + ly, xo = tx0 + lx;
    if (yo < So && xo < So) {
        const int hw = yo * So + xo;
        float* orow = x + (((int64_t)b * To + to) * So * So + hw) * 96 + cg * 24;
        const float* ps = pos_s + (int64_t)hw * 96 + cg * 24;
        const float* pt = pos_t + (int64_t)to * 96 + cg * 24;
#pragma unroll
        for (int e = 0; e < 24; e += 4) {
            const float4 bb = load4(bias + cg * 24 + e), p1 = load4(ps + e), p2 = load4(pt + e);
            // reference order: (conv + bias) + (pos_s + pos_t)
            float4 v;
            v.x = (acc[e] + bb.x) + (p1.x + p2.x);
            v.y = (acc[e + 1] + bb.y) + (p1.y + p2.y);
            v.z = (acc[e + 2] + bb.z) + (p1.z + p2.z);
            v.w = (acc[e + 3] + bb.w) + (p1.w + p2.w);
            store4(orow + e, v);
        }
    }
}


// ------------------------------------------------------------------------------------------------
// bf16 MFMA path: persistent implicit GEMM.
//   K ordering k = (ci*3+dt)*7+dy) * 8 + dx  (63 tap rows + 1 zero row, dx padded 7 -> 8): K = 512 = 32
//   k-steps of 16; the A fragment of a token is then 8 CONSECUTIVE input pixels of one (ci,dt,dy) row --
//   two ds_read_b64 from a bf16 halo patch in LDS, no im2col buffer.
//   One workgroup per CU walks 1x8x16-token tiles.  Waves 0-2 each own one 32-channel block with its
//   whole 32x512 weight slab resident in 128 VGPRs (B fragments) and compute 128 tokens x 32 channels;
//   all four waves share the conversion of the NEXT tile's fp32 halo patch to bf16 into the other LDS buffer: the loads are
//   requested before the MFMA work and converted after it.  Output rows are written 128 B (32 fp32 channels) per half-wave with
//   bias + separable position embedding fused.
// ------------------------------------------------------------------------------------------------
#define SM_TY 8
#define SM_TX 16
#define SM_PH 35                       // 4*8+3 patch rows
#define SM_ROWB 144                    // 72 bf16 per patch row (68 used)
#define SM_PLANE (SM_PH * SM_ROWB)     // 5040
#define SM_PATCH (9 * SM_PLANE)        // 45360

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f4a __attribute__((ext_vector_type(4)));      // 16-byte aligned
#ifndef STEM_ABL
#define STEM_ABL 0     // timing ablations (tools): 1 no halo fill, 2 no MFMA loop, 4 no epilogue stores
#endif

__global__ __launch_bounds__(256, 1) void stem_mfma_kernel(const float* __restrict__ clip, const float* __restrict__ w,
                                                           const float* __restrict__ bias, const float* __restrict__ pos_s,
                                                           const float* __restrict__ pos_t, float* __restrict__ x, int B,
                                                           int T, int S, int To, int So, int tiles_x, int tiles_y) {
    __shared__ __attribute__((aligned(16))) char patch[2][SM_PATCH];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int tiles_per_frame = tiles_x * tiles_y;
    const int ntiles = B * To * tiles_per_frame;

    // ---- B fragments (compute waves): W[n = 32*wave + r][row63 = 2s+h][dx = j] ---------------------
    bf16x8 bfrag[32];
    if (wave < 3) {
        const float* wn = w + (int64_t)(32 * wave + r) * 441;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const int row = 2 * s + h;
            uint32_t pk[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float a = (row < 63) ? wn[row * 7 + 2 * jj] : 0.f;
                const float b2 = (row < 63 && 2 * jj + 1 < 7) ? wn[row * 7 + 2 * jj + 1] : 0.f;
                pk[jj] = pack_bf16x2(a, b2);
            }
            uint4 u = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            bfrag[s] = *reinterpret_cast<bf16x8*>(&u);
        }
    }

    // Halo patch of a tile: 9 (ci,dt) planes x 35 rows x 68 px fp32 -> bf16 in LDS.  Every wave takes every 4th group of 3 rows
    // (27 16-byte loads per lane).  The loads of the NEXT tile are requested before the MFMA work of the current one and
    // converted / written to the other LDS buffer after it, so their latency hides under the matrix work.  Every load is an
    // unconditional 16-byte load at a CLAMPED address (row and column), fixed up afterwards: with bounds tests around the loads
    // the compiler waited for each one (and a single loader wave's 105 round trips per tile were 80 % of the kernel).
    constexpr int FN = 27;
    // 16-byte ALIGNED loads (a 16-byte load at a 4-byte-aligned address is split by the memory pipeline and was 4x slower): a lane
    // loads image pixels [xa, xa+3], xa = 4*tx0 - 4 + 4*seg for 18 segments per row; the LDS piece j holds pixels xa_j+1 .. xa_j+4
    // (patch pixel 0 = image pixel 4*tx0 - 3, which keeps the MFMA waves' 8-byte fragment reads aligned), i.e. elements 1..3 of the
    // lane's quad and element 0 of the next lane's.  Rows and columns outside the image load a clamped address and are zeroed.
    const bool lane_on = lane < 54;
    const int rig = lane_on ? lane / 18 : 0, seg = lane_on ? lane - rig * 18 : 0;    // idle lanes shadow lane 0 (valid addresses)
    auto tile_pos = [&](int tile, int& b, int& to, int& ty0, int& tx0) {
        b = tile / (To * tiles_per_frame);
        int rem = tile - b * (To * tiles_per_frame);
        to = rem / tiles_per_frame;
        rem -= to * tiles_per_frame;
        ty0 = (rem / tiles_x) * SM_TY;
        tx0 = (rem % tiles_x) * SM_TX;
    };
    auto fill_issue = [&](int tile, f4u (&v)[FN], uint32_t& rmask) {
        int b, to, ty0, tx0;
        tile_pos(tile, b, to, ty0, tx0);
        const int xa = 4 * tx0 - 4 + 4 * seg;
        const bool xok = xa >= 0 && xa + 3 < S;
        const int xc = xok ? xa : 0;
        rmask = 0;
#pragma unroll
        for (int u = 0; u < FN; ++u) {
            int rgi = 4 * u + wave;
            rgi = rgi < 105 ? rgi : 104;
            const int row = 3 * rgi + rig;
            const int p = row / SM_PH, py = row - p * SM_PH;
            const int ci = p / 3, dt = p - ci * 3;
            const int ti = 2 * to + dt - 1, yi = 4 * ty0 - 3 + py;
            if (xok && ti >= 0 && ti < T && yi >= 0 && yi < S) rmask |= 1u << u;
            const int tc = ti < 0 ? 0 : (ti >= T ? T - 1 : ti), yc = yi < 0 ? 0 : (yi >= S ? S - 1 : yi);
            v[u] = *reinterpret_cast<const f4a*>(clip + ((((int64_t)b * 3 + ci) * T + tc) * S + yc) * S + xc);
        }
    };
    auto fill_commit = [&](int tile, f4u (&v)[FN], uint32_t rmask, char* dst) {
#pragma unroll
        for (int u = 0; u < FN; ++u) {
            const int rgi = 4 * u + wave;
            const int row = 3 * rgi + rig;
            const float keep = ((rmask >> u) & 1) ? 1.f : 0.f;
            const float f0 = v[u][0] * keep, f1 = v[u][1] * keep, f2 = v[u][2] * keep, f3 = v[u][3] * keep;
            const float nx0 = __shfl_down(f0, 1, 64);          // first pixel of the next segment of the same row
            if (lane_on && seg < 17 && rgi < 105) {
                uint2 o;
                o.x = pack_bf16x2(f1, f2);
                o.y = pack_bf16x2(f3, nx0);
                *reinterpret_cast<uint2*>(dst + row * SM_ROWB + 8 * seg) = o;
            }
        }
    };

    // A-fragment lane base: token r of m-block mb -> (yo_l = 2*mb + (r>>4), xo_l = r&15)
    const int a_base = (4 * (r >> 4)) * SM_ROWB + 8 * (r & 15);
    const int hx = h * (SM_PLANE - 7 * SM_ROWB);   // extra offset when row 2s+1 starts the next (ci,dt) plane
    const float bias_v = (wave < 3) ? bias[32 * wave + r] : 0.f;

    int tile = blockIdx.x;
    f4u fv[FN];
    uint32_t fmask = 0;
    if (!(STEM_ABL & 1) && tile < ntiles) {
        fill_issue(tile, fv, fmask);
        fill_commit(tile, fv, fmask, patch[0]);
    }
    __syncthreads();
    int buf = 0;
    for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        const bool more = !(STEM_ABL & 1) && tile + (int)gridDim.x < ntiles;
        if (more) fill_issue(tile + gridDim.x, fv, fmask);      // lands under the MFMA work below
        if (wave < 3) {
            const char* pb = patch[buf] + a_base;
            f32x16 acc[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mb][i] = 0.f;
#pragma unroll
            for (int s = 0; s < ((STEM_ABL & 2) ? 1 : 32); ++s) {
                const int row0 = 2 * s;                       // tap row of lane-half 0
                const int c0 = (row0 / 7) * SM_PLANE + (row0 % 7) * SM_ROWB;
                int off = c0;
                if (s < 31) {
                    off += h * SM_ROWB;
                    if (row0 % 7 == 6) off += hx;
                }                                             // s == 31: row 63 has zero weights, read row 62 again
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    const char* ap = pb + off + mb * (8 * SM_ROWB);
                    const uint2 lo = *reinterpret_cast<const uint2*>(ap);
                    const uint2 hi = *reinterpret_cast<const uint2*>(ap + 8);
                    uint4 u = make_uint4(lo.x, lo.y, hi.x, hi.y);
                    acc[mb] = mfma16(*reinterpret_cast<bf16x8*>(&u), bfrag[s], acc[mb]);
                }
            }
            // epilogue: + bias + pos_spatial[hw] + pos_temporal[t], token-major fp32
            const int b = tile / (To * tiles_per_frame);
            int rem = tile - b * (To * tiles_per_frame);
            const int to = rem / tiles_per_frame;
            rem -= to * tiles_per_frame;
            const int ty0 = (rem / tiles_x) * SM_TY, tx0 = (rem % tiles_x) * SM_TX;
            const int c = 32 * wave + r;
            const float add_t = bias_v, pt = pos_t[to * 96 + c];
            // the 16 position-embedding loads of an m-block are all requested (at clamped token positions) before the first add:
            // a bounds test around each load made the compiler wait for every single one (64 round trips per tile, 4x the tile's MFMA time)
            float* xt = x + ((int64_t)b * To + to) * So * So * 96 + c;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                float ps[16];
                int hwv[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int tl = (i & 3) + 8 * (i >> 2) + 4 * h;      // token within the m-block
                    const int yo = ty0 + 2 * mb + (tl >> 4), xo = tx0 + (tl & 15);
                    const bool ok = yo < So && xo < So;
                    const int hw = ok ? yo * So + xo : 0;
                    hwv[i] = ok ? hw : -1;
                    ps[i] = pos_s[(int64_t)hw * 96 + c];
                }
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (hwv[i] >= ((STEM_ABL & 4) ? 1 << 30 : 0)) xt[(int64_t)hwv[i] * 96] = (acc[mb][i] + add_t) + (ps[i] + pt);
            }
        }
        if (more) fill_commit(tile + gridDim.x, fv, fmask, patch[buf ^ 1]);
        __syncthreads();
    }
}

extern "C" int mvit_stem_fwd(const float* clip, const float* w, const float* bias, const float* pos_spatial,
                             const float* pos_temporal, float* x, int B, int T, int S, int act_dtype, void* stream) {
    if (!clip || !w || !bias || !pos_spatial || !pos_temporal || !x || B <= 0 || T <= 0 || S <= 0) return MVIT_EINVAL;
    if ((T & 1) || (S & 3)) return MVIT_EUNSUPPORTED;
    const int To = T / 2, So = S / 4;
    if (act_dtype == MVIT_BF16) {
        const int tiles_x = (So + SM_TX - 1) / SM_TX, tiles_y = (So + SM_TY - 1) / SM_TY;
        const int ntiles = B * To * tiles_x * tiles_y;
        const int grid = ntiles < 256 ? ntiles : 256;
        hipLaunchKernelGGL(stem_mfma_kernel, dim3(grid), dim3(256), 0, as_stream(stream), clip, w, bias, pos_spatial,
                           pos_temporal, x, B, T, S, To, So, tiles_x, tiles_y);
        MVIT_LAUNCH_CHECK();
        return MVIT_OK;
    }
    const int tiles = ((So + ST_TX - 1) / ST_TX) * ((So + ST_TY - 1) / ST_TY);
    dim3 grid(tiles, To, B);
    hipLaunchKernelGGL(stem_f32_kernel, grid, dim3(256), 0, as_stream(stream), clip, w, bias, pos_spatial, pos_temporal,
                       x, T, S, To, So);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}
