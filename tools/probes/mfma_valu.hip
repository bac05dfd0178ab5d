// Probe: how much VALU work hides under a bf16 32x32x16 MFMA at ONE wave per SIMD, with the accumulators in arch VGPRs versus in
// ACC registers?  Loop body: 4 MFMAs (4 accumulators), each followed by NF independent v_fma_f32 (8 chains) [+ NE v_exp_f32].
// Reports cycles per MFMA (s_memtime).  Not part of the product.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int N> struct IC {};
template <int MODE, int NF, int NE>      // MODE 0: acc VGPR; 1: acc AGPR; 2: acc VGPR, A and B in AGPR; 3: acc AGPR, A in AGPR
__global__ __launch_bounds__(256, 1) void probe(const uint4* __restrict__ src, float* __restrict__ out, uint64_t* __restrict__ clk, int iters) {
    const int tid = threadIdx.x;
    const uint4 ua = src[tid], ub = src[tid + 256];
    bf16x8 a = *reinterpret_cast<const bf16x8*>(&ua), b = *reinterpret_cast<const bf16x8*>(&ub);
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k) for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
    constexpr bool AGPR = MODE == 1 || MODE == 3;
    asm volatile("v_accvgpr_write_b32 a0, 0" ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","a64","a65","a66","a67","a68","a69","a70","a71");
    asm volatile("v_accvgpr_write_b32 a64, %0\n\tv_accvgpr_write_b32 a65, %1\n\tv_accvgpr_write_b32 a66, %2\n\tv_accvgpr_write_b32 a67, %3\n\tv_accvgpr_write_b32 a68, %4\n\tv_accvgpr_write_b32 a69, %5\n\tv_accvgpr_write_b32 a70, %6\n\tv_accvgpr_write_b32 a71, %7\n\ts_nop 4" :: "v"(ua.x), "v"(ua.y), "v"(ua.z), "v"(ua.w), "v"(ub.x), "v"(ub.y), "v"(ub.z), "v"(ub.w));
    float f[8], e[4];
    for (int i = 0; i < 8; ++i) f[i] = 1.0f + tid * 1e-3f + i;
    for (int i = 0; i < 4; ++i) e[i] = 0.5f + i;
    const float c1 = 0.999f, c2 = 1e-4f;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (MODE == 1) {
                if (k == 0) asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]" ::"v"(a), "v"(b));
                if (k == 1) asm volatile("v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]" ::"v"(a), "v"(b));
                if (k == 2) asm volatile("v_mfma_f32_32x32x16_bf16 a[32:47], %0, %1, a[32:47]" ::"v"(a), "v"(b));
                if (k == 3) asm volatile("v_mfma_f32_32x32x16_bf16 a[48:63], %0, %1, a[48:63]" ::"v"(a), "v"(b));
            } else if (MODE == 3) {
                if (k == 0) asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], a[64:67], %0, a[0:15]" ::"v"(b));
                if (k == 1) asm volatile("v_mfma_f32_32x32x16_bf16 a[16:31], a[64:67], %0, a[16:31]" ::"v"(b));
                if (k == 2) asm volatile("v_mfma_f32_32x32x16_bf16 a[32:47], a[64:67], %0, a[32:47]" ::"v"(b));
                if (k == 3) asm volatile("v_mfma_f32_32x32x16_bf16 a[48:63], a[64:67], %0, a[48:63]" ::"v"(b));
            } else if (MODE == 2) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[64:67], a[68:71], %0" : "+v"(acc[k]));
            } else {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b));
            }
#pragma unroll
            for (int j = 0; j < NF; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[(NF * k + j) & 7]) : "v"(c1), "v"(c2));
#pragma unroll
            for (int j = 0; j < NE; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(e[(NE * k + j) & 3]));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    if (AGPR) { asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a0" : "=v"(s)); }
    else for (int k = 0; k < 4; ++k) for (int i = 0; i < 16; ++i) s += acc[k][i];
    for (int i = 0; i < 8; ++i) s += f[i];
    for (int i = 0; i < 4; ++i) s += e[i];
    out[(size_t)blockIdx.x * 256 + tid] = s;
    if (tid == 0) clk[blockIdx.x] = t1 - t0;
}

template <int AGPR, int NE>
static int go(int nf, int blocks, const uint4* s, float* out, uint64_t* clk, int iters, hipStream_t st) {
#define C(N) case N: probe<AGPR, N, NE><<<blocks, 256, 0, st>>>(s, out, clk, iters); break;
    switch (nf) { C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(10) C(12) default: return -1; }
#undef C
    return (int)hipGetLastError();
}
extern "C" int mfma_valu_launch(int mode, int nf, int ne, int blocks, const void* src, float* out, uint64_t* clk, int iters, hipStream_t st) {
    const uint4* s = (const uint4*)src;
#define M(MD) if (mode == MD) return ne == 0 ? go<MD, 0>(nf, blocks, s, out, clk, iters, st) : ne == 1 ? go<MD, 1>(nf, blocks, s, out, clk, iters, st) : go<MD, 2>(nf, blocks, s, out, clk, iters, st);
    M(0) M(1) M(2) M(3)
#undef M
    return -1;
}
