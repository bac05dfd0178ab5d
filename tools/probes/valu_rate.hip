// VALU issue-rate probe: cycles per instruction of one wave running a chain-free block of 32 independent instructions of one kind,
// with 1 / 2 / 4 waves per SIMD (answers: do v_dot2_f32_bf16 / v_dot2_f32_f16 / v_perm_b32 / v_pk_fma_f32 issue at the v_fma_f32 rate?).
#include <hip/hip_runtime.h>
#include <stdint.h>

#define REP32(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) \
                 X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31)

template <int OP>
__global__ __launch_bounds__(1024) void valu_rate_kernel(const uint32_t* __restrict__ src, float* __restrict__ out, int64_t* __restrict__ clk,
                                                         int iters) {
    const int lane = threadIdx.x & 63;
    uint32_t a = src[lane], b = src[64 + lane];
    float acc[32];
    typedef __attribute__((ext_vector_type(2))) float f2;
    f2 pacc[32], pa = {__uint_as_float(a & 0xffff0000u), __uint_as_float(a << 16)}, pb = {__uint_as_float(b & 0xffff0000u), __uint_as_float(b << 16)};
#pragma unroll
    for (int i = 0; i < 32; ++i) { acc[i] = 0.f; pacc[i] = f2{0.f, 0.f}; }
    const float fa = __uint_as_float(a & 0xffff0000u), fb = __uint_as_float(b << 16);
    const int64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (OP == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(fa), "v"(fb));
            REP32(X)
#undef X
        } else if constexpr (OP == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pacc[i]) : "v"(pa), "v"(pb));
            REP32(X)
#undef X
        } else if constexpr (OP == 2) {
#define X(i) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            REP32(X)
#undef X
        } else if constexpr (OP == 3) {
#define X(i) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            REP32(X)
#undef X
        } else if constexpr (OP == 4) {
#define X(i) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
            REP32(X)
#undef X
        } else if constexpr (OP == 5) {
#define X(i) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(acc[i]) : "v"(a), "v"(b), "v"(0x07060302u));
            REP32(X)
#undef X
        } else if constexpr (OP == 6) {
#define X(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(fa), "v"(fb));
            REP32(X)
#undef X
        } else if constexpr (OP == 7) {
#define X(i) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(acc[i]) : "v"(a));
            REP32(X)
#undef X
        } else if constexpr (OP == 8) {
#define X(i) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
            REP32(X)
#undef X
        }
    }
    const int64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc[i] + pacc[i].x + pacc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

extern "C" int valu_rate_launch(int op, int waves_per_simd, const void* src, void* out, void* clk, int iters, void* stream) {
    dim3 g(256), b(256 * waves_per_simd);
    hipStream_t st = (hipStream_t)stream;
#define L(OP) case OP: hipLaunchKernelGGL((valu_rate_kernel<OP>), g, b, 0, st, (const uint32_t*)src, (float*)out, (int64_t*)clk, iters); break;
    switch (op) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) default: return 1; }
#undef L
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
