// Probe: does the clock the chip holds under MFMA load depend on the MFMA shape?  A bare loop of bf16 MFMAs on Gaussian or zero
// operands, 64x64 output tile per wave, operands either resident in registers or re-read from LDS (ds_read_b128) every k-step.
//   shape 0: v_mfma_f32_32x32x16_bf16 (2x2 accumulators of 16 regs), shape 1: v_mfma_f32_16x16x32_bf16 (4x4 accumulators of 4 regs)
// Reports wall time (host, events) and in-kernel cycles / realtime (clock).  Not part of the product.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE, bool LDS>
__global__ __launch_bounds__(512) void mfma_probe(const uint4* __restrict__ src, float* __restrict__ out, uint64_t* __restrict__ clk, int iters) {
    __shared__ uint4 img[4096];                      // 64 KiB of operand fragments
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += blockDim.x) img[i] = src[i];
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if constexpr (SHAPE == 0) {
        f32x16 acc[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        bf16x8 a[2], b[2];
        for (int i = 0; i < 2; ++i) { a[i] = *reinterpret_cast<bf16x8*>(&img[(wave * 4 + i) * 64 + lane]); b[i] = *reinterpret_cast<bf16x8*>(&img[(wave * 4 + 2 + i) * 64 + lane]); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if constexpr (LDS) {
                    const int base = ((it + ks) & 7) * 512 + lane;        // 8 k-slices of 4 fragments of 64 lanes
#pragma unroll
                    for (int i = 0; i < 2; ++i) { a[i] = *reinterpret_cast<bf16x8*>(&img[base + i * 64]); b[i] = *reinterpret_cast<bf16x8*>(&img[base + 128 + i * 64]); }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
        out[(size_t)blockIdx.x * blockDim.x + tid] = s;
    } else {
        f32x4 acc[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
        bf16x8 a[4], b[4];
        for (int i = 0; i < 4; ++i) { a[i] = *reinterpret_cast<bf16x8*>(&img[(wave * 8 + i) * 64 + lane]); b[i] = *reinterpret_cast<bf16x8*>(&img[(wave * 8 + 4 + i) * 64 + lane]); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {                              // 4 k-steps of 32 = the 8 k-steps of 16 above
                if constexpr (LDS) {
                    const int base = ((it + ks) & 7) * 512 + lane;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { a[i] = *reinterpret_cast<bf16x8*>(&img[base + i * 64]); b[i] = *reinterpret_cast<bf16x8*>(&img[base + 256 + i * 64]); }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) s += acc[i][j][e];
        out[(size_t)blockIdx.x * blockDim.x + tid] = s;
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

extern "C" int mfma_probe_launch(int shape, int lds, int threads, int blocks, const void* src, float* out, uint64_t* clk, int iters, hipStream_t st) {
    const uint4* s = (const uint4*)src;
    if (shape == 0 && lds) mfma_probe<0, true><<<blocks, threads, 0, st>>>(s, out, clk, iters);
    else if (shape == 0) mfma_probe<0, false><<<blocks, threads, 0, st>>>(s, out, clk, iters);
    else if (lds) mfma_probe<1, true><<<blocks, threads, 0, st>>>(s, out, clk, iters);
    else mfma_probe<1, false><<<blocks, threads, 0, st>>>(s, out, clk, iters);
    return (int)hipGetLastError();
}
