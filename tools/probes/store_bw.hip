// Chip-scale store bandwidth by access pattern: a [M][N] 16-bit matrix written once in 128 x 192 tiles (4 waves x (64 x 96)), the way the GEMM
// epilogues do (pattern 0: a lane owns a row, 16-byte pieces, 32 rows x 32 B per wave-instruction) against tile-linear orders (1: the wave's
// 64 x 96 sub-tile in 1-KiB runs = 5.33 rows x 192 B per instruction; 2: the workgroup's 128 x 192 tile in 1-KiB runs = 2.67 rows x 384 B).
#include <hip/hip_runtime.h>
#include <stdint.h>
// patterns 3 / 4: the two-output epilogue of the training fc1 (GELU + saved derivative): pattern 0 into TWO matrices alternately (3: two separate
// buffers, 4: the two outputs side by side in one [M][2N] buffer).
extern "C" __global__ __launch_bounds__(256) void store_bw_kernel(char* __restrict__ out, int64_t M, int N, int pattern) {
    const int ntn = N / 192;
    const int64_t tile = blockIdx.x;
    const int64_t m0 = (tile / ntn) * 128;
    const int n0 = (int)(tile % ntn) * 192;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    const int64_t pitch = (int64_t)N * 2;
    const uint4 v = make_uint4(lane, wave, (uint32_t)tile, 1);
    if (pattern == 0) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                for (int q = 0; q < 4; q += 2)
                    *reinterpret_cast<uint4*>(out + (m0 + 64 * wm + 32 * mb + r) * pitch + (n0 + 96 * wn + 32 * nb + 8 * (q + h)) * 2) = v;
    } else if (pattern == 3 || pattern == 4) {
        char* out2 = pattern == 3 ? out + M * pitch + 4096 * 37 : out + pitch;
        const int64_t p2 = pattern == 3 ? pitch : 2 * pitch;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                for (int q = 0; q < 4; q += 2) {
                    const int64_t o = (m0 + 64 * wm + 32 * mb + r) * p2 + (n0 + 96 * wn + 32 * nb + 8 * (q + h)) * 2;
                    *reinterpret_cast<uint4*>(out + o) = v;
                    *reinterpret_cast<uint4*>(out2 + o) = v;
                }
    } else if (pattern == 5 || pattern == 6) {     // two separate matrices: 5 = all stores of the first, then all of the second; 6 = per 32 x 32 block
        char* out2 = out + M * pitch + 4096 * 37;
        if (pattern == 5) {
#pragma unroll
            for (int which = 0; which < 2; ++which)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                        for (int q = 0; q < 4; q += 2)
                            *reinterpret_cast<uint4*>((which ? out2 : out) + (m0 + 64 * wm + 32 * mb + r) * pitch + (n0 + 96 * wn + 32 * nb + 8 * (q + h)) * 2) = v;
        } else {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                    for (int which = 0; which < 2; ++which)
#pragma unroll
                        for (int q = 0; q < 4; q += 2)
                            *reinterpret_cast<uint4*>((which ? out2 : out) + (m0 + 64 * wm + 32 * mb + r) * pitch + (n0 + 96 * wn + 32 * nb + 8 * (q + h)) * 2) = v;
        }
    } else if (pattern == 1) {
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const int off = j * 1024 + lane * 16, row = off / 192, cb = off - row * 192;
            *reinterpret_cast<uint4*>(out + (m0 + 64 * wm + row) * pitch + (n0 + 96 * wn) * 2 + cb) = v;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const int off = (12 * wave + j) * 1024 + lane * 16, row = off / 384, cb = off - row * 384;
            *reinterpret_cast<uint4*>(out + (m0 + row) * pitch + n0 * 2 + cb) = v;
        }
    }
}
extern "C" int store_bw_launch(void* out, int64_t M, int N, int pattern, void* stream) {
    const int64_t nwg = (M / 128) * (N / 192);
    hipLaunchKernelGGL(store_bw_kernel, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, (char*)out, M, N, pattern);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
