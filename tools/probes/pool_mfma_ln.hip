// Stride-1 pooling conv + LayerNorm on the matrix cores (attention_pool, conv variant: slowfast/models/attention.py:12-83 with the
// depthwise Conv3d(96, 96, 3x3x3, pad 1, groups 96) of :172-212 and LayerNorm(96, eps 1e-5) of :185,199,213) -- im2col on the fly.
//   D^T[16 channels][16 positions] += A[16 ch][K = 32] . B[K][16 pos],  K = 2 taps x 16 channels (v_mfma_f32_16x16x32):
//   A = the two taps' weights on the channel diagonal: constant per (wave, k-step), 14 fragments in registers (27 taps + one zero tap);
//   B = lane (position n, kq): 8 consecutive channels of position p_n + tap(kq >> 1), channel half kq & 1 -- ONE ds_read_b128 from the
//       token-major LDS image [plane][row][position][224 B] (192 B of channels + 32 B pad: with this pitch and this kq order every 16-lane
//       group of a read hits 16 different 16-byte bank groups; tools/probes/pool_mfma_banks.py enumerates the alternatives).
// Useful MACs per MFMA are 6 % -- irrelevant: the family was fp32-VALU / latency-bound (132 issued instructions per 54 FMAs,
// profiles/r2_pmc_pool_*.txt); here the stage-3 q pool is 1.05 M MFMA = ~8 us of matrix pipe and as much LDS time.
// Workgroup = one (batch, head) x (4 rows x 28 columns) of output, marching over T with a ring of 3 input planes (6 rows x 30
// positions incl. halo; the incoming plane waits in registers and takes the slot of the plane that died); 12 waves = 6 channel
// groups of 16 x 2 row pairs; a wave computes 4 (16 positions x 16 channels) blocks per plane: x blocks [0,16) and [12,28) of its two
// rows (28 = 16 + 12: four columns are computed twice, identically).  LayerNorm: per-position partial sums of a wave's 16 channels
// (two cross-lane adds) -> LDS [position][group] -> after the step's barrier every lane reads its position's six partials and
// normalises its own 4 channels in registers.  Conv weights are rounded to the 16-bit type (the VALU kernels keep them fp32):
// covered by the op tests' tolerance and by the model's logit gate.
// STATUS (round 4): built as a product kernel (routed from mvit_pool_conv_ln_fwd_train for stride 1, W % 28 == 0, H % 4 == 0), parity-green
// against the pooling tests' oracle -- and SLOWER than the VALU kernels it was to replace: 71.7 vs 57.5 us (stage 3), 139.8 vs 108.0 (56 x 56),
// 284.6 vs 181.1 (112 x 112), fp16, B = 8 (profiles/r4_pool_mfma_probe.txt).  The conv alone was 34.7 us in the probe (pool_mfma.hip); the
// LayerNorm across six waves (LDS exchange + ds_bpermute adds), two barriers per plane, the 3-slot ring and 29-46 spilled registers at 12
// waves per CU cost more than the conv saved.  Taken out of the library; kept here as the record.  Builds against csrc/common.h:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I aicity_action_amd/csrc -c tools/probes/pool_mfma_ln.hip
#include "common.h"

#define PM_XT 28
#define PM_YT 4
#define PM_XW 30
#define PM_PITCH 224
#define PM_ROWS (PM_YT + 2)
#define PM_PLANE (PM_ROWS * PM_XW * PM_PITCH)        // 40,320 B
#define PM_NT 768
#ifndef PM_DEPTH
#define PM_DEPTH 5
#endif
#define PM_STATS (3 * PM_PLANE)                      // float2 [112 positions][6 groups]
#define PM_SMEM (PM_STATS + PM_YT * PM_XT * 6 * 8)

__device__ __forceinline__ f32x4 pm_mfma(bf16x8 a, bf16x8 b, f32x4 c) {
#ifdef MVIT_HALF_IS_FP16
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mfma16_t, a), __builtin_bit_cast(mfma16_t, b), c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma16_t, a), __builtin_bit_cast(mfma16_t, b), c, 0, 0, 0);
#endif
}

template <int I, int N, typename F>
__device__ __forceinline__ void pm_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        pm_for<I + 1, N>(f);
    }
}
template <int N> using PIC = std::integral_constant<int, N>;

// qkv: [B][T*H*W][ld] 16 bit, the slice of (batch, head) starts at column chan_off + head * 96; w fp32 [96][27]; out / xhat [B][heads][T*H*W][96]
// 16 bit; rstd fp32 [B*heads][T*H*W] (xhat, rstd: NULL in inference)
template <bool TRAIN>
__global__ __launch_bounds__(PM_NT, 1) void pool_mfma_kernel(const bf16_t* __restrict__ qkv, int64_t ld, int chan_off, int heads,
                                                             const float* __restrict__ w, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, bf16_t* __restrict__ out, bf16_t* __restrict__ xhat,
                                                             float* __restrict__ rstd, int T, int H, int W, float eps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n = lane & 15, kq = lane >> 4, kqhi = kq >> 1;
    const int g = wave % 6, hf = wave / 6;
    const int xtiles = W / PM_XT;
    const int ytile = blockIdx.x / xtiles, xtile = blockIdx.x - ytile * xtiles, bh = blockIdx.y;
    const int b = bh / heads, hd = bh - b * heads;
    const int y0 = ytile * PM_YT, x00 = xtile * PM_XT;
    const bf16_t* src = qkv + (int64_t)b * T * H * W * ld + chan_off + hd * 96;

    // ---- weight fragments: k-step j = taps (2j, 2j+1); lane (m = n, kq): A[m][8 kq + e] = (8 (kq & 1) + e == m) ? w[16 g + m][tap] : 0 ----
    bf16x8 af[14];
#pragma unroll
    for (int j = 0; j < 14; ++j) {
        const int tap = 2 * j + kqhi;
        const float wv = tap < 27 ? w[(16 * g + n) * 27 + tap] : 0.f;
        const int e_hit = n - 8 * (kq & 1);
        uint32_t u[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) u[q] = pack_bf16x2(e_hit == 2 * q ? wv : 0.f, e_hit == 2 * q + 1 ? wv : 0.f);
        const uint4 v = make_uint4(u[0], u[1], u[2], u[3]);
        af[j] = *reinterpret_cast<const bf16x8*>(&v);
    }
    const float4 gm = *reinterpret_cast<const float4*>(gamma + 16 * g + 4 * kq), bt = *reinterpret_cast<const float4*>(beta + 16 * g + 4 * kq);

    // staging of input plane tt: rows y0-1 .. y0+4, columns x00-1 .. x00+28, 12 pieces of 16 B each; out-of-image pieces are zeros
    auto stage_load = [&](int tt, uint4 (&buf)[3]) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int pc = tid + PM_NT * i;                 // 6 rows x 30 x 12 = 2160 pieces
            buf[i] = make_uint4(0, 0, 0, 0);
            if (pc < PM_ROWS * PM_XW * 12 && tt >= 0 && tt < T) {
                const int row = pc / (PM_XW * 12), rem = pc - row * (PM_XW * 12), xx = rem / 12, ch = rem - xx * 12;
                const int y = y0 - 1 + row, x = x00 - 1 + xx;
                if (y >= 0 && y < H && x >= 0 && x < W) buf[i] = *reinterpret_cast<const uint4*>(src + ((int64_t)(tt * H + y) * W + x) * ld + 8 * ch);
            }
        }
    };
    auto stage_store = [&](int tt, const uint4 (&buf)[3]) {
        char* pl = smem + ((tt + 3) % 3) * PM_PLANE;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int pc = tid + PM_NT * i;
            if (pc < PM_ROWS * PM_XW * 12) {
                const int pos = pc / 12, ch = pc - pos * 12;
                *reinterpret_cast<uint4*>(pl + pos * PM_PITCH + 16 * ch) = buf[i];
            }
        }
    };
    {
        uint4 buf[3];
        stage_load(-1, buf); stage_store(-1, buf);
        stage_load(0, buf); stage_store(0, buf);
        stage_load(1, buf); stage_store(1, buf);
    }
    __syncthreads();

    // lane address of block u = (row rb of the wave's pair, x block) inside a plane: LDS position (2 hf + rb + dy, x0 + n + dx)
    uint32_t base[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int rb = u >> 1, x0 = (u & 1) * 12;
        base[u] = (uint32_t)(((2 * hf + rb) * PM_XW + x0 + n) * PM_PITCH + (2 * g + (kq & 1)) * 16);
    }
    // B address of (block u, k-step j) = base[u] + plane(tap) + (dy * 30 + dx) * pitch with the lane's tap = 2 j + (kq >> 1).  The even tap's
    // in-plane offset is a compile-time immediate of the read; the odd tap is one position further (dx + 1) or, where dx wraps, 28 positions
    // (next row): two lane-constant deltas; k-step 4 (taps 8 | 9) straddles two planes, k-step 13 pairs tap 26 with the zero tap (same address).
    const uint32_t d_x = kqhi ? PM_PITCH : 0u, d_row = kqhi ? 28u * PM_PITCH : 0u;
    // per-lane element offsets of the four blocks' outputs inside the step's plane (constant over t), validity against H / W
    int eo[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int yl = 2 * hf + (u >> 1), xl = (u & 1) * 12 + n;
        ok[u] = (y0 + yl < H) && (x00 + xl < W);
        eo[u] = (yl * W + xl) * 96 + 16 * g + 4 * kq;
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)(smem);
    float2* const stats = reinterpret_cast<float2*>(smem + PM_STATS);
    const int64_t HW = (int64_t)H * W;

    for (int t = 0; t < T; ++t) {
        uint4 nbuf[3];
        stage_load(t + 2, nbuf);                               // plane t+2: requested before the matrix work, stored behind it
        uint32_t poff[3];
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) poff[dt] = lds0 + (uint32_t)(((t + dt - 1 + 3) % 3) * PM_PLANE);
        // k-step 4: tap 8 = (plane t-1, dy 2, dx 2), tap 9 = (plane t, dy 0, dx 0)
        const uint32_t p4 = kqhi ? poff[1] : poff[0] + (uint32_t)((2 * PM_XW + 2) * PM_PITCH);
        auto bread = [&](bf16x8& dst, auto U_, auto J_) {
            constexpr int u = U_, j = J_;
            if constexpr (j == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(base[u] + p4));
            else {
                constexpr int t0 = 2 * j, dt = t0 / 9, dy = (t0 / 3) % 3, dx = t0 % 3;
                constexpr int imm = (dy * PM_XW + dx) * PM_PITCH;                          // <= 13,888: fits the read's 16-bit offset field
                const uint32_t a = base[u] + poff[dt] + (j == 13 ? 0u : (dx == 2 ? d_row : d_x));
                asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(dst) : "v"(a), "i"(imm));
            }
        };
        // rolling pipeline over the 56 (block, k-step) pairs of the step: PM_DEPTH reads in flight, read i + PM_DEPTH is issued behind MFMA i
        f32x4 acc[4];
        bf16x8 bfr[PM_DEPTH];
        pm_for<0, PM_DEPTH>([&](auto J_) { bread(bfr[J_], PIC<0>{}, J_); });
        pm_for<0, 56>([&](auto I_) {
            constexpr int i = I_, u = i / 14, j = i % 14;
            if constexpr (j == 0) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (i + PM_DEPTH <= 56) asm volatile("s_waitcnt lgkmcnt(%c1)" : "+v"(bfr[i % PM_DEPTH]) : "i"(PM_DEPTH - 1));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bfr[i % PM_DEPTH]));
            acc[u] = pm_mfma(af[j], bfr[i % PM_DEPTH], acc[u]);
            if constexpr (i + PM_DEPTH < 56) bread(bfr[i % PM_DEPTH], PIC<(i + PM_DEPTH) / 14>{}, PIC<(i + PM_DEPTH) % 14>{});
        });
        // LayerNorm statistics: lane (n, kq) holds channels 16 g + 4 kq .. + 3 of position n of each block; the wave's 16 channels = 4 lanes
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float s1 = (acc[u][0] + acc[u][1]) + (acc[u][2] + acc[u][3]);
            float s2 = fmaf(acc[u][0], acc[u][0], fmaf(acc[u][1], acc[u][1], fmaf(acc[u][2], acc[u][2], acc[u][3] * acc[u][3])));
            s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
            s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
            if (kq == 0) stats[((2 * hf + (u >> 1)) * PM_XT + (u & 1) * 12 + n) * 6 + g] = make_float2(s1, s2);
        }
        __syncthreads();                                       // the step's matrix work is done everywhere: plane t-1 is dead, the partial sums are visible
        stage_store(t + 2, nbuf);
        const int64_t pbase = ((int64_t)bh * T + t) * HW + (int64_t)y0 * W + x00;      // (scalar) first position of the tile in this plane
        bf16_t* const op = out + pbase * 96;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4* sp = reinterpret_cast<const float4*>(stats + ((2 * hf + (u >> 1)) * PM_XT + (u & 1) * 12 + n) * 6);
            const float4 p0 = sp[0], p1 = sp[1], p2 = sp[2];
            const float S1 = ((p0.x + p0.z) + (p1.x + p1.z)) + (p2.x + p2.z), S2 = ((p0.y + p0.w) + (p1.y + p1.w)) + (p2.y + p2.w);
            const float mean = S1 * (1.0f / 96.0f);
            const float var = fmaxf(S2 * (1.0f / 96.0f) - mean * mean, 0.f);
            const float rs = 1.0f / sqrtf(var + eps);
            if (ok[u]) {
                const float h0 = (acc[u][0] - mean) * rs, h1 = (acc[u][1] - mean) * rs, h2 = (acc[u][2] - mean) * rs, h3 = (acc[u][3] - mean) * rs;
                *reinterpret_cast<uint2*>(op + eo[u]) =
                    make_uint2(pack_bf16x2(fmaf(h0, gm.x, bt.x), fmaf(h1, gm.y, bt.y)), pack_bf16x2(fmaf(h2, gm.z, bt.z), fmaf(h3, gm.w, bt.w)));
                if (TRAIN) {
                    *reinterpret_cast<uint2*>(xhat + pbase * 96 + eo[u]) = make_uint2(pack_bf16x2(h0, h1), pack_bf16x2(h2, h3));
                    if (g == 0 && kq == 0) rstd[pbase + (eo[u] - 16 * g - 4 * kq) / 96] = rs;
                }
            }
        }
        __syncthreads();                                       // plane t+2 is in place; the partial sums may be overwritten
    }
}

int mvit_internal_pool_mfma_fwd(const void* qkv, int64_t ld, int chan_off, const float* w, const float* gamma, const float* beta, void* out,
                                void* xhat, float* rstd, int B, int heads, int T, int H, int W, float eps, hipStream_t st) {
    if (W % PM_XT != 0 || H % PM_YT != 0 || (ld & 7) || (chan_off & 7)) return MVIT_EUNSUPPORTED;
    static DevFlags attr_tab;
    bool& attr_done = dev_flag(attr_tab);
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pool_mfma_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, PM_SMEM) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&pool_mfma_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, PM_SMEM) != hipSuccess)
            return MVIT_ELAUNCH;
        attr_done = true;
    }
    dim3 grid((W / PM_XT) * (H / PM_YT), B * heads);
    if (xhat)
        hipLaunchKernelGGL((pool_mfma_kernel<true>), grid, dim3(PM_NT), PM_SMEM, st, (const bf16_t*)qkv, ld, chan_off, heads, w, gamma, beta, (bf16_t*)out,
                           (bf16_t*)xhat, rstd, T, H, W, eps);
    else
        hipLaunchKernelGGL((pool_mfma_kernel<false>), grid, dim3(PM_NT), PM_SMEM, st, (const bf16_t*)qkv, ld, chan_off, heads, w, gamma, beta, (bf16_t*)out,
                           nullptr, nullptr, T, H, W, eps);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}
