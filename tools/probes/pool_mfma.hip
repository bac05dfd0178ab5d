// Probe (round 4, review item 2): the stride-1 pooling conv of a stage-3 block on the matrix cores -- im2col on the fly.
//   attention_pool, conv variant (slowfast/models/attention.py:12-83): depthwise Conv3d(96, 96, 3x3x3, pad 1, groups 96) on the
//   token-major q slice of qkv [B][T*H*W][ld] (16 bit), here WITHOUT the LayerNorm that follows (the probe times the conv's new form
//   and checks it against conv3d; LayerNorm is ~6 VALU per element on whole rows).
// D^T[16 channels][16 positions] += A[16 ch][32] . B[32][16 pos], K = 2 taps x 16 channels:
//   A = the two taps' weights on the channel diagonal -- constant per (wave, k-step), 14 fragments in registers;
//   B = lane (n = position, kq): 8 consecutive channels of position p_n + tap(kq >> 1), channel half kq & 1: ONE ds_read_b128 from the
//       LDS image [plane][row][position][224 B] (192 B of channels + 32 B pad: with this pitch and this kq order every 16-lane group of a
//       read hits 16 different 16-byte bank groups -- tools/probes/pool_mfma_banks.py).
// Workgroup = one (batch, head) x 4 output rows, marching over T with a ring of 4 input planes (6 rows x 30 positions incl. halo);
// 12 waves = 6 channel groups x 2 row pairs; a wave computes 4 (16 positions x 16 channels) blocks per plane: x blocks [0,16) and
// [12,28) of its two rows (28 = 16 + 12: the overlap is recomputed).  Staging is plain global loads + ds_write (unoptimised).
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 mfma16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define PM_W 28
#define PM_T 8
#define PM_YT 4
#define PM_XW 30
#define PM_PITCH 224
#define PM_ROWS (PM_YT + 2)
#define PM_PLANE (PM_ROWS * PM_XW * PM_PITCH)        // 40,320 B
#define PM_NT 768

__device__ __forceinline__ uint32_t pm_pack(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f2;
    typedef __attribute__((ext_vector_type(2))) __bf16 h2;
    const f2 f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, h2));
}

// qkv: [BH / heads][T*H*W][ld] 16 bit, the slice of (batch, head) bh starts at column col0 + (bh % heads) * 96; w fp32 [96][27]; out [BH][T*H*W][96] 16 bit
__global__ __launch_bounds__(PM_NT, 1) void pool_mfma_probe_kernel(const uint16_t* __restrict__ qkv, int64_t ld, int col0, int heads,
                                                                    const float* __restrict__ w, uint16_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n = lane & 15, kq = lane >> 4;
    const int g = wave % 6, hf = wave / 6;
    const int ytile = blockIdx.x, bh = blockIdx.y;
    const int b = bh / heads, hd = bh - b * heads;
    const int y0 = ytile * PM_YT;
    const uint16_t* src = qkv + (int64_t)b * (PM_T * PM_W * PM_W) * ld + col0 + hd * 96;

    // ---- weight fragments: k-step j = taps (2j, 2j+1); lane (m = n, kq): A[m][8 kq + e] = (8 (kq & 1) + e == m) ? w[16 g + m][tap] : 0 ----
    bf16x8 af[14];
#pragma unroll
    for (int j = 0; j < 14; ++j) {
        const int tap = 2 * j + (kq >> 1);
        const float wv = tap < 27 ? w[(16 * g + n) * 27 + tap] : 0.f;
        const int e_hit = n - 8 * (kq & 1);
        uint32_t u[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) u[q] = pm_pack(e_hit == 2 * q ? wv : 0.f, e_hit == 2 * q + 1 ? wv : 0.f);
        const uint4 v = make_uint4(u[0], u[1], u[2], u[3]);
        af[j] = *reinterpret_cast<const bf16x8*>(&v);
    }
    // ---- zero the ring once (halo columns and out-of-image rows / planes stay zero) -------------------------------------------------
    for (int i = tid; i < 4 * PM_PLANE / 16; i += PM_NT) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    // staging of input plane tt into ring slot tt & 3: rows y0-1 .. y0+4, 28 positions x 12 pieces of 16 B
    auto stage_load = [&](int tt, uint4 (&buf)[3]) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int pc = tid + PM_NT * i;                 // 6 rows x 28 x 12 = 2016 pieces
            buf[i] = make_uint4(0, 0, 0, 0);
            if (pc < PM_ROWS * PM_W * 12 && tt >= 0 && tt < PM_T) {
                const int row = pc / (PM_W * 12), rem = pc - row * (PM_W * 12), x = rem / 12, ch = rem - x * 12;
                const int y = y0 - 1 + row;
                if (y >= 0 && y < PM_W) buf[i] = *reinterpret_cast<const uint4*>(src + ((int64_t)(tt * PM_W + y) * PM_W + x) * ld + 8 * ch);
            }
        }
    };
    auto stage_store = [&](int tt, const uint4 (&buf)[3]) {
        char* pl = smem + ((tt + 4) & 3) * PM_PLANE;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int pc = tid + PM_NT * i;
            if (pc < PM_ROWS * PM_W * 12) {
                const int row = pc / (PM_W * 12), rem = pc - row * (PM_W * 12), x = rem / 12, ch = rem - x * 12;
                *reinterpret_cast<uint4*>(pl + (row * PM_XW + x + 1) * PM_PITCH + 16 * ch) = buf[i];
            }
        }
    };
    {
        uint4 buf[3];
        stage_load(0, buf); stage_store(0, buf);
        stage_load(1, buf); stage_store(1, buf);
    }
    __syncthreads();

    // lane address of (row block rb, x block xb) inside a plane: position (1 + 2 hf + rb + dy - 1, x0 + n + dx) -> tap (dy, dx) adds (dy * 30 + dx) * pitch
    uint32_t base[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int rb = u >> 1, x0 = (u & 1) * 12;
        base[u] = (uint32_t)(((2 * hf + rb) * PM_XW + x0 + n) * PM_PITCH + (2 * g + (kq & 1)) * 16);
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)(smem);
    for (int t = 0; t < PM_T; ++t) {
        uint4 nbuf[3];
#ifdef PM_ABL_NOSTAGE
        nbuf[0] = nbuf[1] = nbuf[2] = make_uint4(0, 0, 0, 0);
        if (t > 100)
#endif
        stage_load(t + 2, nbuf);                               // plane t+2 -> slot of plane t-2 (free): requested before the matrix work
        uint32_t toff[14];
#pragma unroll
        for (int j = 0; j < 14; ++j) {
            int tap = 2 * j + (kq >> 1);
            tap = tap < 27 ? tap : 13;                       // (the zero tap reads the centre: its weights are zero)
            const int dt = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
            toff[j] = lds0 + (uint32_t)(((t + dt - 1 + 4) & 3) * PM_PLANE + (dy * PM_XW + dx) * PM_PITCH);
        }
        // rolling pipeline over the 56 (block, k-step) pairs of the step: 7 reads in flight, read i + 7 is issued behind MFMA i
        f32x4 acc[4];
        bf16x8 bfr[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(bfr[j]) : "v"(base[0] + toff[j]));
#pragma unroll
        for (int i = 0; i < 56; ++i) {
            const int u = i / 14, j = i % 14;
            if (j == 0) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (i + 7 <= 56) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(bfr[i % 7]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bfr[i % 7]));
#ifdef PM_ABL_NOMFMA
            acc[u][0] += __builtin_bit_cast(float, (int)bfr[i % 7][0]);
#else
            acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma16_t, af[j]), __builtin_bit_cast(mfma16_t, bfr[i % 7]), acc[u], 0, 0, 0);
#endif
            if (i + 7 < 56) asm volatile("ds_read_b128 %0, %1" : "=v"(bfr[i % 7]) : "v"(base[(i + 7) / 14] + toff[(i + 7) % 14]));
        }
        // D^T: lane (n = position, mq = kq) holds channels 16 g + 4 mq .. + 3 of position n
#ifdef PM_ABL_NOSTORE
        if (acc[0][0] == 12345.f)
#endif
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rb = u >> 1, x0 = (u & 1) * 12;
            const int y = y0 + 2 * hf + rb, x = x0 + n;
            uint2 v = make_uint2(pm_pack(acc[u][0], acc[u][1]), pm_pack(acc[u][2], acc[u][3]));
            *reinterpret_cast<uint2*>(out + (((int64_t)bh * PM_T + t) * PM_W * PM_W + y * PM_W + x) * 96 + 16 * g + 4 * kq) = v;
        }
        stage_store(t + 2, nbuf);                             // (slot (t+2) & 3 held plane t-2: free since the barrier that closed step t-1)
        __syncthreads();
    }
}

extern "C" int pool_mfma_probe_launch(const void* qkv, int64_t ld, int col0, int heads, const void* w, void* out, int BH, void* stream) {
    static bool done = false;
    if (!done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pool_mfma_probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * PM_PLANE) != hipSuccess) return -3;
        done = true;
    }
    hipLaunchKernelGGL(pool_mfma_probe_kernel, dim3(PM_W / PM_YT, BH), dim3(PM_NT), 4 * PM_PLANE, (hipStream_t)stream, (const uint16_t*)qkv, ld, col0, heads,
                       (const float*)w, (uint16_t*)out);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
