// Shared device helpers for the gfx950 MViT kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>

#include "../../include/mvit_hip.h"

// The 16-bit activation type.  One source tree builds two libraries: libmvit_hip.so (bfloat16, the default) and, with
// -DMVIT_HALF_IS_FP16, libmvit_hip_f16.so in which every "bf16" below means IEEE half (same MFMA rate on gfx950, 3 more
// mantissa bits -> closes the 1e-3 logit gate that bf16 storage cannot).  All bit-level conversions go through the
// helpers of this header; `bf16_t` stays the name of the raw 16-bit storage type in both builds.
typedef uint16_t bf16_t;  // raw 16-bit storage (bf16, or fp16 under MVIT_HALF_IS_FP16)

typedef __attribute__((ext_vector_type(8))) short bf16x8;    // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;   // 32x32 accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define MVIT_LAUNCH_CHECK()                                   \
    do {                                                      \
        if (hipGetLastError() != hipSuccess) return MVIT_ELAUNCH; \
    } while (0)

#ifdef MVIT_HALF_IS_FP16
typedef __attribute__((ext_vector_type(8))) _Float16 mfma16_t;
#define MVIT_ONE16 ((short)0x3C00)
__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
    _Float16 h;
    __builtin_memcpy(&h, &v, 2);
    return (float)h;
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {     // round-to-nearest-even (v_cvt_f16_f32)
    _Float16 h = (_Float16)f;
    bf16_t r;
    __builtin_memcpy(&r, &h, 2);
    return r;
}
__device__ __forceinline__ float lo16_to_f32(uint32_t u) { return bf16_to_f32((bf16_t)(u & 0xffffu)); }
__device__ __forceinline__ float hi16_to_f32(uint32_t u) { return bf16_to_f32((bf16_t)(u >> 16)); }
#else
typedef __attribute__((ext_vector_type(8))) __bf16 mfma16_t;
#define MVIT_ONE16 ((short)0x3F80)
__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
    return __uint_as_float(((uint32_t)v) << 16);
}

// round-to-nearest-even; NaN stays NaN (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return *reinterpret_cast<bf16_t*>(&b);
}
__device__ __forceinline__ float lo16_to_f32(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi16_to_f32(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
#endif

// two fp32 -> one dword of two 16-bit values (lo in bits 0-15): a single v_cvt_pk_* instruction
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_v;
#ifdef MVIT_HALF_IS_FP16
    typedef __attribute__((ext_vector_type(2))) _Float16 h16x2_v;
#else
    typedef __attribute__((ext_vector_type(2))) __bf16 h16x2_v;
#endif
    const f32x2_v f = {lo, hi};
    const h16x2_v v = __builtin_convertvector(f, h16x2_v);
    return __builtin_bit_cast(uint32_t, v);
}

template <typename T> struct ActIO;
template <> struct ActIO<float> {
    static __device__ __forceinline__ float load(const float* p) { return *p; }
    static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct ActIO<bf16_t> {
    static __device__ __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
    static __device__ __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// load 4 consecutive activations as floats (16 B for fp32, 8 B for bf16)
__device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 load4(const bf16_t* p) {
    uint2 u = *reinterpret_cast<const uint2*>(p);
    float4 r;
    r.x = lo16_to_f32(u.x);
    r.y = hi16_to_f32(u.x);
    r.z = lo16_to_f32(u.y);
    r.w = hi16_to_f32(u.y);
    return r;
}
// 8 consecutive bf16 (one 16-byte load) -> two float4
__device__ __forceinline__ void load8(const bf16_t* p, float4& lo, float4& hi) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    lo.x = lo16_to_f32(u.x); lo.y = hi16_to_f32(u.x);
    lo.z = lo16_to_f32(u.y); lo.w = hi16_to_f32(u.y);
    hi.x = lo16_to_f32(u.z); hi.y = hi16_to_f32(u.z);
    hi.z = lo16_to_f32(u.w); hi.w = hi16_to_f32(u.w);
}
__device__ __forceinline__ void load8(const float* p, float4& lo, float4& hi) {
    lo = *reinterpret_cast<const float4*>(p);
    hi = *reinterpret_cast<const float4*>(p + 4);
}
__device__ __forceinline__ void store4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void store4(bf16_t* p, float4 v) {
    uint2 u;
    u.x = pack_bf16x2(v.x, v.y);
    u.y = pack_bf16x2(v.z, v.w);
    *reinterpret_cast<uint2*>(p) = u;
}

__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// Fast erf-GELU for the 16-bit paths.  Phi(x) = 1 / (1 + 2^(x q(x^2))) with x q(x^2) = -log2(e) * p(x), p an odd minimax fit of
// logit(Phi(x)) on |x| <= 5 (the logit of the normal CDF is odd, so one polynomial serves both tails and keeps the RELATIVE accuracy
// of GELU(x) = x Phi(x) in the negative tail): 7 plain VALU + v_exp + v_rcp per element, against 14 + 2 for the Abramowitz-Stegun
// 7.1.26 form used before (the fc1 epilogue was VALU-bound on it: 14 VALU instructions per MFMA, profiles/r2_pmc_gemm_fc1.txt).
//   bf16 build: degree 7, |GELU error| <= 1.3e-3 |GELU| (2^-9.6) for |x| <= 5 and <= 8e-7 beyond, absolute <= 1.7e-4 -- below the
//               2^-9 rounding of the 16-bit result; derivative within 2.7e-4;
//   fp16 build: degree 13, relative <= 1.1e-4, absolute <= 1.2e-5 (the result keeps 11 bits), derivative within 2.3e-5.
// Coefficients and the error scan: tools/fit_gelu.py.  p is increasing on the whole real line, so large |x| saturate to 0 / x.
#ifdef MVIT_HALF_IS_FP16
#define MVIT_GELU_NK 7
__device__ static const float k_gelu[MVIT_GELU_NK] = {-2.301758256e+00f, -1.055243742e-01f, 3.938158157e-04f, 1.041289184e-04f, -6.513817968e-06f, 1.755232626e-07f, -1.842319936e-09f};
#else
#define MVIT_GELU_NK 4
__device__ static const float k_gelu[MVIT_GELU_NK] = {-2.297646723e+00f, -1.093967574e-01f, 1.423557742e-03f, -1.303203199e-05f};
#endif
__device__ __forceinline__ float gelu_phi_fast(float x, float x2) {        // Phi(x); x2 = x * x
    float q = k_gelu[MVIT_GELU_NK - 1];
#pragma unroll
    for (int i = MVIT_GELU_NK - 2; i >= 0; --i) q = fmaf(q, x2, k_gelu[i]);
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(q * x));
}
__device__ __forceinline__ float gelu_fast(float x) { return x * gelu_phi_fast(x, x * x); }
__device__ __forceinline__ void gelu_and_grad_fast(float x, float& gval, float& gder) {     // GELU_erf(x) and its derivative Phi(x) + x phi(x)
    const float x2 = x * x, cdf = gelu_phi_fast(x, x2);
    const float e = __builtin_amdgcn_exp2f(x2 * -0.72134752044448170368f);          // exp(-x^2 / 2)
    gval = x * cdf;
    gder = fmaf(x * 0.39894228040143267794f, e, cdf);
}
__device__ __forceinline__ float gelu_grad_fast(float x) {
    float gv, gd;
    gelu_and_grad_fast(x, gv, gd);
    return gd;
}
// D = A(32x16) * B(16x32) + C on 16-bit operands held as raw bits (bf16x8 = 8 shorts); fp32 accumulate
__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
#ifdef MVIT_HALF_IS_FP16
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(mfma16_t, a), __builtin_bit_cast(mfma16_t, b), c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma16_t, a), __builtin_bit_cast(mfma16_t, b), c, 0, 0, 0);
#endif
}

// XCD-aware (tile, group) mapping for grids with x = tiles of one group and y = groups that share an operand (the K/V of one
// (batch, head), ...).  Workgroups are dispatched round-robin over the 8 XCDs in linear-id order (x fastest), so with the
// identity mapping the tiles of one group land on all 8 L2s and every L2 has to hold every group's operand; here each XCD gets
// whole groups, so a group's operand is fetched into ONE L2 and hit there by all its tiles.
__device__ __forceinline__ void xcd_group_map(int& tile, int& group) {
    const int nt = gridDim.x, ng = gridDim.y;
    tile = blockIdx.x;
    group = blockIdx.y;
    if ((ng & 7) == 0) {
        const int lin = blockIdx.y * nt + blockIdx.x;
        const int xcd = lin & 7, slot = lin >> 3;
        const int gq = slot / nt;
        group = gq * 8 + xcd;
        tile = slot - gq * nt;
    }
}

// A library-owned side stream per device with a fork / join pair of events: independent kernels of one entry point (the dQ and
// the dK/dV pass of the attention backward) are issued on `st` and on the side stream between fork() and join(), so each fills
// the other's partially occupied last wave of workgroups.  After join() everything is ordered on `st` again, so callers see
// ordinary single-stream semantics (buffers may be reused / freed in stream order).  Returns nullptr when disabled
// (MVIT_NO_SIDE_STREAM) or on failure: the caller then launches everything on `st`.
struct SideStream {
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
};
static inline SideStream* side_stream_for_current_device() {
    static SideStream tab[16];
    static const bool off = getenv("MVIT_NO_SIDE_STREAM") != nullptr;
    if (off) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    SideStream* s = &tab[dev];
    if (!s->side) {
        // MVIT_SIDE_PRIO (probe; default 0 = plain stream): 1 = lowest stream priority for the side stream, 2 = highest.  Lowest bought 0.1 ms of the
        // train step on one box and nothing on another -- and a process that had created the prioritised stream ran the three sub-batch
        // streams of the inference forward at 577 instead of 728 clips/s afterwards (the stream -> hardware queue assignment changes:
        // profiles/r5_side_stream_priority_ab.txt).  Not adopted.
        static const int prio_env = getenv("MVIT_SIDE_PRIO") ? atoi(getenv("MVIT_SIDE_PRIO")) : 0;
        int least = 0, greatest = 0;
        if (prio_env && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess) {
            if (hipStreamCreateWithPriority(&s->side, hipStreamNonBlocking, prio_env == 1 ? least : greatest) != hipSuccess) { s->side = nullptr; return nullptr; }
        } else
        if (hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking) != hipSuccess) { s->side = nullptr; return nullptr; }
        if (hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming) != hipSuccess)
            return nullptr;
    }
    return s;
}
static inline bool side_fork(SideStream* s, hipStream_t st) {
    return s && hipEventRecord(s->ev_fork, st) == hipSuccess && hipStreamWaitEvent(s->side, s->ev_fork, 0) == hipSuccess;
}
static inline bool side_join(SideStream* s, hipStream_t st) {
    return s && hipEventRecord(s->ev_join, s->side) == hipSuccess && hipStreamWaitEvent(st, s->ev_join, 0) == hipSuccess;
}

// Lazily set per-kernel state (hipFuncSetAttribute for > 64 KiB of dynamic LDS, the CU count behind a persistent grid) is a property
// of the DEVICE the call runs on, not of the process: one slot per device ordinal.
// Slots are atomics (two host threads driving two devices may race on first use; the worst case is a repeated, idempotent
// hipFuncSetAttribute); an ordinal beyond the table, or a failing hipGetDevice, gets a scratch slot that is never cached -- the
// attribute is then set on every launch instead of aliasing another device's "done" flag.
#include <atomic>
constexpr int MVIT_MAX_DEVS = 64;
struct DevFlag {
    std::atomic<bool>* slot;          // nullptr = the uncached scratch case (reads false, writes dropped)
    operator bool() const { return slot ? slot->load(std::memory_order_acquire) : false; }
    DevFlag& operator=(bool v) { if (slot) slot->store(v, std::memory_order_release); return *this; }
};
struct DevFlags { std::atomic<bool> f[MVIT_MAX_DEVS] = {}; };
static inline int dev_slot() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MVIT_MAX_DEVS) return -1;
    return dev;
}
static inline DevFlag dev_flag(DevFlags& d) {
    const int dev = dev_slot();
    return DevFlag{dev >= 0 ? &d.f[dev] : nullptr};
}
struct DevInt {
    std::atomic<int>* slot;
    operator int() const { return slot ? slot->load(std::memory_order_acquire) : 0; }
    DevInt& operator=(int v) { if (slot) slot->store(v, std::memory_order_release); return *this; }
};
struct DevInts { std::atomic<int> v[MVIT_MAX_DEVS] = {}; };
static inline DevInt dev_int(DevInts& d) {
    const int dev = dev_slot();
    return DevInt{dev >= 0 ? &d.v[dev] : nullptr};
}

// CU count of the current device, cached per device ordinal; -1 on error
static inline int dev_cu_count(DevInts& tab) {
    DevInt c = dev_int(tab);
    int n = c;
    if (n > 0) return n;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
        return -1;
    c = n;
    return n;
}

// An LDS pointer made opaque to the optimiser (keeps loop-invariant reads INSIDE a loop instead of pinning dozens of registers)
// WITHOUT losing its address space: the usual `asm volatile("" : "+v"(ptr))` on a generic pointer cuts the trace back to the
// __shared__ array, and every read through it becomes a FLAT load -- counted in vmcnt AND lgkmcnt and only completed by
// `s_waitcnt vmcnt(0)`, i.e. it also waits for every global load / LDS-DMA / store the wave has in flight (found in the pooling
// kernels in round 5: each frame's weight reads waited for the NEXT frame's prefetch).  Launder the 32-bit LDS offset instead.
#ifdef __HIPCC__
typedef const __attribute__((address_space(3))) char* lds_cptr_t;
typedef float lds_f32x2_t __attribute__((ext_vector_type(2)));       // (HIP's float2 class cannot be copied out of an address-space-3 object)
__device__ __forceinline__ lds_cptr_t lds_opaque(const void* p) {
    uint32_t a = (uint32_t)(uintptr_t)(lds_cptr_t)(const char*)p;
    asm volatile("" : "+v"(a));
    return (lds_cptr_t)(uintptr_t)a;
}
template <typename T>
__device__ __forceinline__ T lds_ld(lds_cptr_t p, int byte_off) {
    return *reinterpret_cast<const __attribute__((address_space(3))) T*>(p + byte_off);
}
#endif

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// deterministic column sums of a [nparts][width] fp32 partial table (backward_rowops.hip)
int mvit_internal_reduce_partials(const float* part, int nparts, int width, float* out_a, float* out_b, int split, int accumulate,
                                  hipStream_t st, int defer_ok = 0);
