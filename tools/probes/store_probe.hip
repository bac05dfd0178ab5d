// Micro-probe: what does one global_store_dwordx4 wave-instruction cost a CU, by access pattern?  (tools only, not product code)
// 256 workgroups x 512 threads; every wave issues NST stores per round of a pattern into its own region; cycles per round by s_memtime.
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" __global__ __launch_bounds__(512, 2) void store_probe(char* out, float* cyc, int pattern, int rounds, int row_stride, int nwaves_active) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= nwaves_active) return;
    char* base = out + ((size_t)blockIdx.x * 8 + wave) * (4u << 20);      // 4 MiB per wave
    uint4 v = make_uint4(lane, wave, blockIdx.x, 1);
    uint64_t t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rounds; ++r) {
        char* p = base + (size_t)(r & 3) * (1u << 20);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            size_t off;
            if (pattern == 0) off = (size_t)((lane & 15) + 16 * (i / 3)) * row_stride + (i % 3) * 64 + (lane >> 4) * 16;   // 16 rows x 64 B (the GEMM epilogue)
            else if (pattern == 1) off = (size_t)i * 1024 + lane * 16;                                                    // 1 KiB contiguous
            else if (pattern == 2) off = (size_t)((lane >> 3) + 8 * i) * row_stride + (lane & 7) * 16;                    // 8 rows x 128 B
            else if (pattern == 3) off = (size_t)(lane + 64 * (i / 6)) * row_stride + (i % 6) * 16;                       // 64 rows x 16 B (row per lane)
            else off = (size_t)((lane >> 2) + 16 * (i / 3)) * row_stride + (i % 3) * 64 + (lane & 3) * 16;                // 16 rows x 64 B, lanes of a row adjacent
            *reinterpret_cast<uint4*>(p + off) = v;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint64_t t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = (float)(t1 - t0) / rounds;
}
