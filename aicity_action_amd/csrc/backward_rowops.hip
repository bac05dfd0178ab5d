// Backward / training-step row kernels (HBM-bound): LayerNorm bwd, GELU fwd/bwd, skip max-pool bwd, column sums
// (bias gradients), soft-target cross entropy, global grad-norm, AdamW.  All reductions are two-stage and
// deterministic (no float atomics) unless noted.
#include <mutex>

#include "common.h"

template <int LPR>
__device__ __forceinline__ float gsum(float v) {
#pragma unroll
    for (int off = LPR / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ----------------------------------------------------------------------------------------------
// LayerNorm backward.  x fp32 [rows][C] (forward input, stats recomputed), dy TDY [rows][C] or, in broadcast mode,
// dy fp32 [rows/rows_per_dy][C] scaled by dy_scale (head: d(mean over tokens)).
// dx fp32 [rows][C] (+= if accumulate).  part: [gridDim.x][2][C] partial dgamma / dbeta.
// ----------------------------------------------------------------------------------------------
template <int C, typename TDY>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const TDY* __restrict__ dy, int64_t rows_per_dy, float dy_scale,
                                                     const float* base, float* dx, float* __restrict__ part,
                                                     int64_t rows, float eps, bf16_t* __restrict__ dx16,
                                                     const float* __restrict__ scale16, int64_t rps16) {
    constexpr int LPR = C / 12;
    constexpr int RPB = 256 / LPR;
    __shared__ float red[RPB][2 * C];
    const int lir = threadIdx.x % LPR, rib = threadIdx.x / LPR;
    float4 g[3], ag[3], ab[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        g[i] = load4(gamma + 4 * (lir + LPR * i));
        ag[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int64_t r0 = (int64_t)blockIdx.x * RPB; r0 < rows; r0 += (int64_t)gridDim.x * RPB) {
        const int64_t r = r0 + rib;
        const bool ok = r < rows;
        float4 v[3], d[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            d[i] = v[i];
            if (ok) {
                v[i] = load4(x + r * C + 4 * (lir + LPR * i));
                d[i] = load4(dy + (r / rows_per_dy) * C + 4 * (lir + LPR * i));
                d[i].x *= dy_scale; d[i].y *= dy_scale; d[i].z *= dy_scale; d[i].w *= dy_scale;
            }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        const float mean = gsum<LPR>(s) * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
            q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
        }
        const float rstd = 1.0f / sqrtf(gsum<LPR>(q) * (1.0f / C) + eps);
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            v[i].x *= rstd; v[i].y *= rstd; v[i].z *= rstd; v[i].w *= rstd;   // xhat
            ag[i].x += d[i].x * v[i].x; ag[i].y += d[i].y * v[i].y; ag[i].z += d[i].z * v[i].z; ag[i].w += d[i].w * v[i].w;
            ab[i].x += d[i].x; ab[i].y += d[i].y; ab[i].z += d[i].z; ab[i].w += d[i].w;
            d[i].x *= g[i].x; d[i].y *= g[i].y; d[i].z *= g[i].z; d[i].w *= g[i].w;   // g*dy
            c1 += (d[i].x + d[i].y) + (d[i].z + d[i].w);
            c2 += (d[i].x * v[i].x + d[i].y * v[i].y) + (d[i].z * v[i].z + d[i].w * v[i].w);
        }
        c1 = gsum<LPR>(c1) * (1.0f / C);
        c2 = gsum<LPR>(c2) * (1.0f / C);
        if (ok) {
            const float sc16 = (dx16 && scale16) ? scale16[r / rps16] : 1.0f;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                float* p = dx + r * C + 4 * (lir + LPR * i);
                float4 o;
                o.x = rstd * (d[i].x - c1 - v[i].x * c2);
                o.y = rstd * (d[i].y - c1 - v[i].y * c2);
                o.z = rstd * (d[i].z - c1 - v[i].z * c2);
                o.w = rstd * (d[i].w - c1 - v[i].w * c2);
                if (base) {          // dx = base + LN-backward (base == dx: in-place accumulate; else e.g. the residual-stream gradient)
                    const float4 old = load4(base + r * C + 4 * (lir + LPR * i));
                    o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
                }
                store4(p, o);
                if (dx16) {          // the same gradient as the 16-bit operand of the next GEMMs (what mvit_cast_rows_f32_to_bf16 would write)
                    o.x *= sc16; o.y *= sc16; o.z *= sc16; o.w *= sc16;
                    store4(dx16 + r * C + 4 * (lir + LPR * i), o);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        *reinterpret_cast<float4*>(&red[rib][4 * (lir + LPR * i)]) = ag[i];
        *reinterpret_cast<float4*>(&red[rib][C + 4 * (lir + LPR * i)]) = ab[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * C; c += 256) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < RPB; ++r) s += red[r][c];
        part[(int64_t)blockIdx.x * 2 * C + c] = s;
    }
}

// out[j] (+)= sum_b part[b][j]: deterministic column sums of a [nparts][width] partial table in ONE launch, no float atomics.
// grid = (64-column blocks, row slices).  Every workgroup sums its slice of the rows (4 interleaved row groups, fixed order)
// and, when there is more than one slice, parks the result in a per-launch scratch row; the slice that ARRIVES LAST for a
// column block (write-through stores + ticket, L1-bypassing loads on the last arriver: cdna guide G16, sc1 form) adds the parked rows
// in slice order -- so the summation order is a function of the launch geometry only, never of timing.  Scratch rows and
// tickets are static device arrays of the library (tickets reset themselves).  Slots belong to STREAMS: every stream that
// reduces gets its own ring of RED_RING slots (red_slot_base below), so two reductions can only share a slot when they are
// ordered on one stream -- a process-wide round-robin counter let a reduction of one sub-batch stream land on the slot of a
// still-pending reduction of the other once the counter wrapped (about 130 reductions per backward chain against 64 slots).
#define RED_SLOTS 128                // 8 stream rings x 16 slots (43 MB of scratch rows)
#define RED_MAXSLICE 32
#define RED_MAXW 2624                 // widest table: 27 x 96 pooling-conv weight gradients (2592), padded to 64
__device__ float g_red_scratch[RED_SLOTS][RED_MAXSLICE][RED_MAXW];
__device__ unsigned g_red_ticket[RED_SLOTS][RED_MAXW / 64];

// one (column block bx, row slice by of nsl) unit of the reduction described above
__device__ __forceinline__ void reduce_partials_body(const float* __restrict__ part, int nparts, int width, float* __restrict__ out_a,
                                                     float* __restrict__ out_b, int split, int accumulate, int slot, int bx, int by,
                                                     int nsl, float (*red)[64], unsigned* last) {
    const int col = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int j = bx * 64 + col;
    const int per = (nparts + nsl - 1) / nsl;
    const int b0 = by * per, b1 = min(nparts, b0 + per);
    float s = 0.f;
    if (j < width)
        for (int b = b0 + sl; b < b1; b += 4) s += part[(int64_t)b * width + j];
    red[sl][col] = s;
    __syncthreads();
    if (sl != 0) {
        if (nsl == 1) return;
    } else {
        s = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
        if (nsl == 1) {
            if (j < width) {
                float* o = (j < split) ? out_a + j : out_b + (j - split);
                *o = accumulate ? *o + s : s;
            }
            return;
        }
        // write-through (sc1) store: visible to every CU without a release fence once this wave's vmcnt has drained
        __hip_atomic_store(&g_red_scratch[slot][by][j < RED_MAXW ? j : 0], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // publish (cdna guide G16, sc1 form): the storing wave drains its stores, the workgroup meets, ONE lane takes a ticket; the
    // last arriver reads the parked rows with sc1 loads (L1 bypassed: no acquire fence needed)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(&g_red_ticket[slot][bx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *last = (t == (unsigned)nsl - 1u) ? 1u : 0u;
        if (*last) __hip_atomic_store(&g_red_ticket[slot][bx], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
    __syncthreads();
    if (!*last || sl != 0 || j >= width) return;
    float tot = 0.f;
    for (int y = 0; y < nsl; ++y) tot += __hip_atomic_load(&g_red_scratch[slot][y][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float* o = (j < split) ? out_a + j : out_b + (j - split);
    *o = accumulate ? *o + tot : tot;
}

__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, int nparts, int width,
                                                              float* __restrict__ out_a, float* __restrict__ out_b,
                                                              int split, int accumulate, int slot) {
    __shared__ float red[4][64];
    __shared__ unsigned last;
    reduce_partials_body(part, nparts, width, out_a, out_b, split, accumulate, slot, blockIdx.x, blockIdx.y, gridDim.y, red, &last);
}

// A batch of such reductions in ONE launch (descriptors by value in the kernel arguments: no upload): the block backward queues
// its eight small reductions (two LayerNorms, three pooling convs x {LayerNorm, conv weights}) and flushes them together --
// 16 launches per step instead of 131, each of which cost ~11 us for a few microseconds of work.
#define RED_QMAX 16
struct RedDesc { const float* part; float* out_a; float* out_b; int nparts, width, split, accumulate, slot, nsl, first; };
struct RedBatch { RedDesc d[RED_QMAX]; int n; };

__global__ __launch_bounds__(256) void reduce_partials_multi_kernel(const RedBatch b) {
    __shared__ float red[4][64];
    __shared__ unsigned last;
    int i = 0;
#pragma unroll 1
    for (int k = 1; k < b.n; ++k) i = ((int)blockIdx.x >= b.d[k].first) ? k : i;
    const RedDesc& d = b.d[i];
    const int u = blockIdx.x - d.first;           // unit = column block * nsl + slice
    reduce_partials_body(d.part, d.nparts, d.width, d.out_a, d.out_b, d.split, d.accumulate, d.slot, u / d.nsl, u % d.nsl, d.nsl, red, &last);
}

static RedBatch g_red_queue;
static bool g_red_queue_on = false;
static int g_red_units = 0;

// stream -> ring of RED_RING scratch slots (per device: the scratch arrays are per-device objects).  A ninth stream on one device
// takes over the least recently used ring after that ring's stream has drained.
#define RED_RING 16
#define RED_NRING (RED_SLOTS / RED_RING)
struct RedRing { hipStream_t st; bool used; unsigned next; unsigned long long stamp; };
static RedRing g_red_rings[16][RED_NRING];
static unsigned long long g_red_stamp = 0;
static std::mutex g_red_mutex;
static int red_slot_base(hipStream_t st, int nslots, int* base) {          // nslots consecutive slots of st's ring (nslots <= RED_RING)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return MVIT_ELAUNCH;
    std::lock_guard<std::mutex> lock(g_red_mutex);
    RedRing* rings = g_red_rings[dev];
    int pick = -1;
    for (int i = 0; i < RED_NRING; ++i)
        if (rings[i].used && rings[i].st == st) { pick = i; break; }
    if (pick < 0) {
        for (int i = 0; i < RED_NRING && pick < 0; ++i)
            if (!rings[i].used) pick = i;
        if (pick < 0) {             // all rings taken: the least recently used one, once its stream has nothing pending
            pick = 0;
            for (int i = 1; i < RED_NRING; ++i)
                if (rings[i].stamp < rings[pick].stamp) pick = i;
            // (a ninth stream: rare -- the main, weight-gradient, library side, sub-batch and capture streams of a process make five to seven)
            if (hipStreamQuery(rings[pick].st) != hipSuccess && hipStreamSynchronize(rings[pick].st) != hipSuccess) return MVIT_ELAUNCH;
        }
        rings[pick].st = st; rings[pick].used = true; rings[pick].next = 0;
    }
    rings[pick].stamp = ++g_red_stamp;
    if (rings[pick].next + (unsigned)nslots > RED_RING) rings[pick].next = 0;       // consecutive, no wrap inside one launch
    *base = pick * RED_RING + (int)rings[pick].next;
    rings[pick].next = (rings[pick].next + (unsigned)nslots) % RED_RING;
    return MVIT_OK;
}

static int red_flush(hipStream_t st) {
    if (g_red_queue.n > 0) {
        int base = 0;
        const int rc = red_slot_base(st, g_red_queue.n, &base);        // slots are taken on the stream the batch RUNS on
        if (rc != MVIT_OK) { g_red_queue.n = 0; g_red_units = 0; return rc; }
        for (int i = 0; i < g_red_queue.n; ++i) g_red_queue.d[i].slot = base + i;
        hipLaunchKernelGGL(reduce_partials_multi_kernel, dim3((unsigned)g_red_units), dim3(256), 0, st, g_red_queue);
        g_red_queue.n = 0;
        g_red_units = 0;
        MVIT_LAUNCH_CHECK();
    }
    return MVIT_OK;
}

// Deferred mode (see include/mvit_hip.h): between _begin and _flush the LayerNorm / pooling-conv backward entry points queue their
// parameter-gradient reductions instead of launching them; _flush(stream) launches the batch on `stream`, which must be ordered
// after every producer.  The partial tables live in the callers' workspaces: those must stay untouched until the flush.
extern "C" int mvit_reduce_queue_begin(void) {
    g_red_queue.n = 0;
    g_red_units = 0;
    g_red_queue_on = true;
    return MVIT_OK;
}
extern "C" int mvit_reduce_queue_flush(void* stream) {
    g_red_queue_on = false;
    return red_flush(as_stream(stream));
}

// shared by every file of the library that reduces a partial table (declared in common.h); defer_ok: this reduction may wait in
// the queue (its partial table is not reused by the caller before the flush)
int mvit_internal_reduce_partials(const float* part, int nparts, int width, float* out_a, float* out_b, int split, int accumulate,
                                  hipStream_t st, int defer_ok) {
    if (width > RED_MAXW) return MVIT_EUNSUPPORTED;
    int slices = nparts / 64;
    slices = slices < 1 ? 1 : (slices > RED_MAXSLICE ? RED_MAXSLICE : slices);
    if (defer_ok && g_red_queue_on) {
        if (g_red_queue.n == RED_QMAX) {            // full: what is queued goes out now, on this stream
            const int rc = red_flush(st);
            if (rc != MVIT_OK) return rc;
        }
        RedDesc& d = g_red_queue.d[g_red_queue.n++];
        d.part = part; d.out_a = out_a; d.out_b = out_b; d.nparts = nparts; d.width = width; d.split = split;
        d.accumulate = accumulate; d.slot = 0; d.nsl = slices; d.first = g_red_units;      // slot: assigned at flush
        g_red_units += ((width + 63) / 64) * slices;
        return MVIT_OK;
    }
    int slot = 0;
    const int rc = red_slot_base(st, 1, &slot);
    if (rc != MVIT_OK) return rc;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((width + 63) / 64, slices), dim3(256), 0, st, part, nparts, width, out_a, out_b, split,
                       accumulate, slot);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}
static int launch_reduce_partials(const float* part, int nparts, int width, float* out_a, float* out_b, int split, int accumulate,
                                  hipStream_t st, int defer_ok = 0) {
    return mvit_internal_reduce_partials(part, nparts, width, out_a, out_b, split, accumulate, st, defer_ok);
}

#define LN_BWD_MAXBLK 1024     // 4 workgroups per CU: the kernel is HBM-bound
extern "C" int64_t mvit_layernorm_bwd_workspace_bytes(int C) { return (int64_t)LN_BWD_MAXBLK * 2 * C * sizeof(float); }

template <int C, typename TDY>
static int launch_ln_bwd(const float* x, const float* gamma, const void* dy, int64_t rpd, float dys, const float* base, float* dx,
                         float* dgamma, float* dbeta, int acc_param, float* ws, int64_t rows, float eps, hipStream_t st,
                         void* dx16, const float* scale16, int64_t rps16) {
    constexpr int RPB = 256 / (C / 12);
    int64_t blocks = (rows + RPB - 1) / RPB;
    if (blocks > LN_BWD_MAXBLK) blocks = LN_BWD_MAXBLK;
    hipLaunchKernelGGL((ln_bwd_kernel<C, TDY>), dim3((unsigned)blocks), dim3(256), 0, st, x, gamma, (const TDY*)dy, rpd, dys,
                       base, dx, ws, rows, eps, (bf16_t*)dx16, scale16, rps16);
    MVIT_LAUNCH_CHECK();
    return launch_reduce_partials(ws, (int)blocks, 2 * C, dgamma, dbeta, C, acc_param, st, 1);
}

// dy_dtype: MVIT_F32 / MVIT_BF16; rows_per_dy > 1 => broadcast mode (dy must be fp32 [rows/rows_per_dy][C]).
// dx = (dx_base ? dx_base : 0) + LayerNorm-backward(dy); dx_base may alias dx (in-place accumulate) or be a different buffer
// (the block backward adds the norm-2 branch onto the incoming stream gradient without cloning it first).
extern "C" int mvit_layernorm_bwd(const float* x, const float* gamma, const void* dy, int dy_dtype, int64_t rows_per_dy,
                                   float dy_scale, const float* dx_base, float* dx, float* dgamma, float* dbeta,
                                   int accumulate_param, float* workspace, int64_t rows, int C, float eps, void* dx16,
                                   const float* dx16_row_scale, int64_t dx16_rows_per_scale, void* stream) {
    if (!x || !gamma || !dy || !dx || !dgamma || !dbeta || !workspace || rows <= 0 || rows_per_dy <= 0) return MVIT_EINVAL;
    if (dx16 && dx16_row_scale && dx16_rows_per_scale <= 0) return MVIT_EINVAL;
    hipStream_t st = as_stream(stream);
#define LNB(CC)                                                                                                      \
    case CC:                                                                                                         \
        if (dy_dtype == MVIT_F32)                                                                                    \
            return launch_ln_bwd<CC, float>(x, gamma, dy, rows_per_dy, dy_scale, dx_base, dx, dgamma, dbeta,          \
                                            accumulate_param, workspace, rows, eps, st, dx16, dx16_row_scale,        \
                                            dx16_rows_per_scale);                                                    \
        if (dy_dtype == MVIT_BF16 && rows_per_dy == 1)                                                               \
            return launch_ln_bwd<CC, bf16_t>(x, gamma, dy, 1, dy_scale, dx_base, dx, dgamma, dbeta,                   \
                                             accumulate_param, workspace, rows, eps, st, dx16, dx16_row_scale,       \
                                             dx16_rows_per_scale);                                                   \
        return MVIT_EDTYPE;
    switch (C) {
        LNB(96) LNB(192) LNB(384) LNB(768)
        default: return MVIT_EUNSUPPORTED;
    }
#undef LNB
}

// ----------------------------------------------------------------------------------------------
// GELU (exact erf) forward / backward, elementwise, n % 4 == 0.
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
    return cdf + x * pdf;
}

template <typename T>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 v = load4(x + 4 * i);
        v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w);
        store4(y + 4 * i, v);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
                                                       int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = load4(x + 4 * i);
        float4 d = load4(dy + 4 * i);
        if constexpr (sizeof(T) == 2) {
            d.x *= gelu_grad_fast(v.x); d.y *= gelu_grad_fast(v.y); d.z *= gelu_grad_fast(v.z); d.w *= gelu_grad_fast(v.w);
        } else {
            d.x *= gelu_grad(v.x); d.y *= gelu_grad(v.y); d.z *= gelu_grad(v.z); d.w *= gelu_grad(v.w);
        }
        store4(dx + 4 * i, d);
    }
}

static unsigned ew_grid(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (unsigned)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

extern "C" int mvit_gelu_fwd(const void* x, void* y, int64_t n, int act_dtype, void* stream) {
    if (!x || !y || n < 0 || (n & 3)) return MVIT_EINVAL;
    if (n == 0) return MVIT_OK;
    if (act_dtype == MVIT_F32)
        hipLaunchKernelGGL((gelu_fwd_kernel<float>), dim3(ew_grid(n / 4)), dim3(256), 0, as_stream(stream), (const float*)x, (float*)y, n / 4);
    else if (act_dtype == MVIT_BF16)
        hipLaunchKernelGGL((gelu_fwd_kernel<bf16_t>), dim3(ew_grid(n / 4)), dim3(256), 0, as_stream(stream), (const bf16_t*)x, (bf16_t*)y, n / 4);
    else
        return MVIT_EDTYPE;
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}

extern "C" int mvit_gelu_bwd(const void* x, const void* dy, void* dx, int64_t n, int act_dtype, void* stream) {
    if (!x || !dy || !dx || n < 0 || (n & 3)) return MVIT_EINVAL;
    if (n == 0) return MVIT_OK;
    if (act_dtype == MVIT_F32)
        hipLaunchKernelGGL((gelu_bwd_kernel<float>), dim3(ew_grid(n / 4)), dim3(256), 0, as_stream(stream), (const float*)x, (const float*)dy, (float*)dx, n / 4);
    else if (act_dtype == MVIT_BF16)
        hipLaunchKernelGGL((gelu_bwd_kernel<bf16_t>), dim3(ew_grid(n / 4)), dim3(256), 0, as_stream(stream), (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, n / 4);
    else
        return MVIT_EDTYPE;
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}

// ----------------------------------------------------------------------------------------------
// Skip max-pool backward (gather form, deterministic): dx[in] = sum over the <=4 windows that contain `in` and
// whose FIRST maximum (scan order dy,dx as ATen) is `in`.
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_skip_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               float* __restrict__ dx, int BT, int H, int W, int Ho, int Wo,
                                                               int C4) {
    const int64_t total = (int64_t)BT * H * W * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        int64_t tok = i / C4;
        const int xi = (int)(tok % W); tok /= W;
        const int yi = (int)(tok % H);
        const int64_t bt = tok / H;
        const float4 me = load4(x + i * 4);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // windows (yo,xo) with 2*yo-1 <= yi <= 2*yo+1
        for (int yo = (yi) / 2; yo <= (yi + 1) / 2; ++yo) {
            if (yo < 0 || yo >= Ho) continue;
            for (int xo = (xi) / 2; xo <= (xi + 1) / 2; ++xo) {
                if (xo < 0 || xo >= Wo) continue;
                // is `me` the first maximum of this window, per channel?
                float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
                int4 bidx = make_int4(-1, -1, -1, -1);
                int myidx = -1;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int yy = 2 * yo + ky - 1;
                    if (yy < 0 || yy >= H) continue;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int xx = 2 * xo + kx - 1;
                        if (xx < 0 || xx >= W) continue;
                        const int widx = ky * 3 + kx;
                        if (yy == yi && xx == xi) myidx = widx;
                        const float4 v = load4(x + (((bt * H + yy) * W + xx) * C4 + c4) * 4);
                        if (v.x > best.x) { best.x = v.x; bidx.x = widx; }
                        if (v.y > best.y) { best.y = v.y; bidx.y = widx; }
                        if (v.z > best.z) { best.z = v.z; bidx.z = widx; }
                        if (v.w > best.w) { best.w = v.w; bidx.w = widx; }
                    }
                }
                const float4 g = load4(dy + (((bt * Ho + yo) * Wo + xo) * C4 + c4) * 4);
                if (bidx.x == myidx) acc.x += g.x;
                if (bidx.y == myidx) acc.y += g.y;
                if (bidx.z == myidx) acc.z += g.z;
                if (bidx.w == myidx) acc.w += g.w;
            }
        }
        (void)me;
        store4(dx + i * 4, acc);
    }
}

extern "C" int mvit_maxpool_skip_bwd(const float* x, const float* dy, float* dx, int B, int T, int H, int W, int C,
                                     void* stream) {
    if (!x || !dy || !dx || B <= 0 || T <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return MVIT_EINVAL;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int64_t total = (int64_t)B * T * H * W * (C / 4);
    hipLaunchKernelGGL(maxpool_skip_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, as_stream(stream), x, dy, dx, B * T, H, W,
                       Ho, Wo, C / 4);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}

// ----------------------------------------------------------------------------------------------
// Training form of the skip max-pool: the forward also records, per output element, which of the 9 window positions
// held the FIRST maximum (ATen scan order ky,kx); the backward then reads one index byte + one gradient per window an
// input belongs to (1, 2 or 4 windows) instead of re-scanning the windows.
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_skip_idx_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               uint32_t* __restrict__ idx, int BT, int H, int W, int Ho, int Wo, int C4) {
    const int64_t total = (int64_t)BT * Ho * Wo * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        int64_t tok = i / C4;
        const int xo = (int)(tok % Wo); tok /= Wo;
        const int yo = (int)(tok % Ho);
        const int64_t bt = tok / Ho;
        // nine unconditional loads at clamped coordinates (all in flight together); out-of-frame taps are then excluded from the
        // arg-max scan by the validity mask, so the recorded position is the first maximum among the in-frame taps (ATen order)
        float4 v[9];
        bool ok9[9];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yi0 = 2 * yo + ky - 1;
            const int yi = yi0 < 0 ? 0 : (yi0 >= H ? H - 1 : yi0);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xi0 = 2 * xo + kx - 1;
                const int xi = xi0 < 0 ? 0 : (xi0 >= W ? W - 1 : xi0);
                ok9[ky * 3 + kx] = yi0 == yi && xi0 == xi;
                v[ky * 3 + kx] = load4(x + (((bt * H + yi) * W + xi) * C4 + c4) * 4);
            }
        }
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        uint32_t bx = 0, by = 0, bz = 0, bw = 0;
#pragma unroll
        for (uint32_t w = 0; w < 9; ++w) {
            if (ok9[w] && v[w].x > m.x) { m.x = v[w].x; bx = w; }
            if (ok9[w] && v[w].y > m.y) { m.y = v[w].y; by = w; }
            if (ok9[w] && v[w].z > m.z) { m.z = v[w].z; bz = w; }
            if (ok9[w] && v[w].w > m.w) { m.w = v[w].w; bw = w; }
        }
        store4(y + i * 4, m);
        idx[i] = bx | (by << 8) | (bz << 16) | (bw << 24);
    }
}

__global__ __launch_bounds__(256) void maxpool_skip_bwd_idx_kernel(const uint32_t* __restrict__ idx, const float* __restrict__ dy,
                                                                   float* __restrict__ dx, int BT, int H, int W, int Ho, int Wo, int C4) {
    const int64_t total = (int64_t)BT * H * W * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        int64_t tok = i / C4;
        const int xi = (int)(tok % W); tok /= W;
        const int yi = (int)(tok % H);
        const int64_t bt = tok / H;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int yo = yi / 2; yo <= (yi + 1) / 2; ++yo) {
            if (yo >= Ho) continue;
            for (int xo = xi / 2; xo <= (xi + 1) / 2; ++xo) {
                if (xo >= Wo) continue;
                const uint32_t me = (uint32_t)((yi - 2 * yo + 1) * 3 + (xi - 2 * xo + 1));
                const int64_t o = ((bt * Ho + yo) * Wo + xo) * C4 + c4;
                const uint32_t w = idx[o];
                const float4 g = load4(dy + o * 4);
                if ((w & 255u) == me) acc.x += g.x;
                if (((w >> 8) & 255u) == me) acc.y += g.y;
                if (((w >> 16) & 255u) == me) acc.z += g.z;
                if ((w >> 24) == me) acc.w += g.w;
            }
        }
        store4(dx + i * 4, acc);
    }
}

extern "C" int mvit_maxpool_skip_fwd_idx(const float* x, float* y, void* idx, int B, int T, int H, int W, int C, void* stream) {
    if (!x || !y || !idx || B <= 0 || T <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return MVIT_EINVAL;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int64_t total = (int64_t)B * T * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(maxpool_skip_idx_kernel, dim3(ew_grid(total)), dim3(256), 0, as_stream(stream), x, y, (uint32_t*)idx, B * T, H, W,
                       Ho, Wo, C / 4);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}

extern "C" int mvit_maxpool_skip_bwd_idx(const void* idx, const float* dy, float* dx, int B, int T, int H, int W, int C, void* stream) {
    if (!idx || !dy || !dx || B <= 0 || T <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return MVIT_EINVAL;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int64_t total = (int64_t)B * T * H * W * (C / 4);
    hipLaunchKernelGGL(maxpool_skip_bwd_idx_kernel, dim3(ew_grid(total)), dim3(256), 0, as_stream(stream), (const uint32_t*)idx, dy, dx,
                       B * T, H, W, Ho, Wo, C / 4);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}

// ----------------------------------------------------------------------------------------------
// Column sums: out[n] (+)= scale-weighted sum over rows of a[M][N]  (bias gradients; two-stage).
// ----------------------------------------------------------------------------------------------
#define CS_MAXBLK 512      // 2 workgroups per CU (HBM-bound); 512 x 96 floats is also what the pooling backward's workspace holds
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ a, int64_t M, int N,
                                                             const float* __restrict__ row_scale, int64_t rps,
                                                             float* __restrict__ part) {
    // 2-D mapping: tpr = min(N/4, 256) threads per row (4 columns each), 256/tpr rows per pass; LDS combine of the row slots
    extern __shared__ float red[];   // [rows_per_pass][N]  (only when rows_per_pass > 1)
    const int n4 = N / 4;
    const int tpr = n4 < 256 ? n4 : 256;
    const int rpp = 256 / tpr;
    const int rslot = threadIdx.x / tpr, c0 = threadIdx.x % tpr;
    const bool on = rslot < rpp;
    for (int c = c0; c < n4; c += tpr) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (on)
            for (int64_t m = (int64_t)blockIdx.x * rpp + rslot; m < M; m += (int64_t)gridDim.x * rpp) {
                const float4 v = load4(a + m * N + 4 * c);
                const float sc = row_scale ? row_scale[m / rps] : 1.f;
                s.x += sc * v.x; s.y += sc * v.y; s.z += sc * v.z; s.w += sc * v.w;
            }
        if (rpp == 1) {
            if (on) *reinterpret_cast<float4*>(part + (int64_t)blockIdx.x * N + 4 * c) = s;
        } else {
            if (on) *reinterpret_cast<float4*>(red + rslot * N + 4 * c) = s;
            __syncthreads();
            if (rslot == 0) {
                for (int r = 1; r < rpp; ++r) {
                    const float4 t = *reinterpret_cast<const float4*>(red + r * N + 4 * c);
                    s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
                }
                *reinterpret_cast<float4*>(part + (int64_t)blockIdx.x * N + 4 * c) = s;
            }
            __syncthreads();
        }
    }
}

extern "C" int64_t mvit_colsum_workspace_bytes(int N) { return (int64_t)CS_MAXBLK * N * sizeof(float); }

extern "C" int mvit_colsum(const void* a, int a_dtype, int64_t M, int N, const float* row_scale, int64_t rows_per_scale,
                           float* out, int accumulate, float* workspace, void* stream) {
    if (!a || !out || !workspace || M <= 0 || N <= 0 || (N & 3)) return MVIT_EINVAL;
    hipStream_t st = as_stream(stream);
    const int blocks = (int)(M < CS_MAXBLK ? M : CS_MAXBLK);
    const int n4 = N / 4, tpr = n4 < 256 ? n4 : 256, rpp = 256 / tpr;
    const size_t lds = rpp > 1 ? (size_t)rpp * N * sizeof(float) : 0;
    if (lds > 60000) return MVIT_EUNSUPPORTED;
    if (a_dtype == MVIT_F32)
        hipLaunchKernelGGL((colsum_partial_kernel<float>), dim3(blocks), dim3(256), lds, st, (const float*)a, M, N, row_scale, rows_per_scale, workspace);
    else if (a_dtype == MVIT_BF16)
        hipLaunchKernelGGL((colsum_partial_kernel<bf16_t>), dim3(blocks), dim3(256), lds, st, (const bf16_t*)a, M, N, row_scale, rows_per_scale, workspace);
    else
        return MVIT_EDTYPE;
    MVIT_LAUNCH_CHECK();
    return launch_reduce_partials(workspace, blocks, N, out, out, N, accumulate, st);
}

// ----------------------------------------------------------------------------------------------
// Soft-target cross entropy (slowfast/models/losses.py:133-142, reduction mean) forward + dlogits, B x ncls small.
// ----------------------------------------------------------------------------------------------
__global__ void softce_kernel(const float* __restrict__ logits, const float* __restrict__ y, float* __restrict__ loss,
                              float* __restrict__ dlogits, int B, int ncls, float grad_scale) {
    extern __shared__ float sm[];
    float* per = sm;  // [B]
    const int b = threadIdx.x;
    if (b < B) {
        const float* l = logits + (int64_t)b * ncls;
        const float* t = y + (int64_t)b * ncls;
        float m = -INFINITY;
        for (int j = 0; j < ncls; ++j) m = fmaxf(m, l[j]);
        float den = 0.f, ysum = 0.f;
        for (int j = 0; j < ncls; ++j) { den += expf(l[j] - m); ysum += t[j]; }
        const float lse = m + logf(den);
        float acc = 0.f;
        for (int j = 0; j < ncls; ++j) {
            acc += -t[j] * (l[j] - lse);
            if (dlogits) dlogits[(int64_t)b * ncls + j] = grad_scale * (expf(l[j] - lse) * ysum - t[j]) / (float)B;
        }
        per[b] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < B; ++i) s += per[i];
        *loss = s / (float)B;
    }
}

extern "C" int mvit_soft_ce(const float* logits, const float* labels, float* loss, float* dlogits, int B, int num_classes,
                            float grad_scale, void* stream) {
    if (!logits || !labels || !loss || B <= 0 || B > 1024 || num_classes <= 0) return MVIT_EINVAL;
    hipLaunchKernelGGL(softce_kernel, dim3(1), dim3(((B + 63) / 64) * 64), B * sizeof(float), as_stream(stream), logits, labels,
                       loss, dlogits, B, num_classes, grad_scale);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}

// ----------------------------------------------------------------------------------------------
// Multi-tensor helpers over a device table of (pointer, count) chunks: squared-norm partials and AdamW.
// The host flattens the parameter list once into chunk descriptors (<= CHUNK elements each).
// ----------------------------------------------------------------------------------------------
struct MtChunk { float* p; float* g; float* m; float* v; int n; float wd; };

__global__ __launch_bounds__(256) void sqnorm_chunks_kernel(const MtChunk* __restrict__ ch, float* __restrict__ part) {
    __shared__ float red[4];
    const MtChunk c = ch[blockIdx.x];
    // 16-byte loads, four independent partial sums per thread (the scalar form read 141 MB of gradients at 1.25 TB/s).  The
    // element -> (thread, accumulator) map is the same whether or not the chunk is 16-byte aligned (gradients that are views into
    // a DistributedDataParallel bucket need not be): the sum, and with it the clip factor, does not depend on where a gradient lives
    const bool al = (reinterpret_cast<uintptr_t>(c.g) & 15) == 0;
    const int n4 = c.n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(c.g);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (al) {
#pragma unroll 4
        for (int i = threadIdx.x; i < n4; i += 256) {
            const float4 g = g4[i];
            s0 += g.x * g.x; s1 += g.y * g.y; s2 += g.z * g.z; s3 += g.w * g.w;
        }
    } else {
        for (int i = threadIdx.x; i < n4; i += 256) {
            const float gx = c.g[4 * i], gy = c.g[4 * i + 1], gz = c.g[4 * i + 2], gw = c.g[4 * i + 3];
            s0 += gx * gx; s1 += gy * gy; s2 += gz * gz; s3 += gw * gw;
        }
    }
    float s = (s0 + s1) + (s2 + s3);
    for (int i = 4 * n4 + threadIdx.x; i < c.n; i += 256) { const float g = c.g[i]; s += g * g; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// total = sqrt(sum part); coef = min(1, max_norm / (total + 1e-6))  (torch clip_grad_norm_ semantics); max_norm<=0: coef=1
__global__ void gradnorm_finish_kernel(const float* __restrict__ part, int n, float max_norm, float* __restrict__ out2) {
    // one wave: strided partial sums in double, then a fixed-order butterfly (a single thread walking ~900 dependent loads took 35 us)
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) s += (double)part[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float tot = (float)sqrt(s);
        out2[0] = tot;
        out2[1] = max_norm > 0.f ? fminf(1.0f, max_norm / (tot + 1e-6f)) : 1.0f;
    }
}

// AdamW, decoupled weight decay, bias correction (torch.optim.AdamW, amsgrad False); grads scaled by *coef (clip).
__global__ __launch_bounds__(256) void adamw_chunks_kernel(const MtChunk* __restrict__ ch, const float* __restrict__ coef_ptr,
                                                           float lr, float beta1, float beta2, float eps, float bc1,
                                                           float bc2_sqrt, const float* __restrict__ hyper) {
    const MtChunk c = ch[blockIdx.x];
    if (hyper) {        // {lr, 1 - beta1^t, sqrt(1 - beta2^t)} from device memory: a captured (hipGraph) step replays with new values
        lr = hyper[0];
        bc1 = hyper[1];
        bc2_sqrt = hyper[2];
    }
    const float coef = coef_ptr ? coef_ptr[1] : 1.0f;
    // a non-finite global norm (NaN / inf loss or gradient) must not reach the parameters or the moments: skip the whole step
    // (bit test: the build uses -fno-honor-nans)
    if (coef_ptr && (((__float_as_uint(coef_ptr[0]) & 0x7f800000u) == 0x7f800000u) || ((__float_as_uint(coef) & 0x7f800000u) == 0x7f800000u)))
        return;
    // every operation spelled out with its rounding (no contraction left to the compiler): the 16-byte path and the scalar path
    // below -- which one runs depends on the alignment of the gradient, e.g. a view into a DistributedDataParallel bucket -- must
    // give the same bits
    const float decay = __fsub_rn(1.0f, __fmul_rn(lr, c.wd)), step = __fdiv_rn(lr, bc1);
    const float omb1 = __fsub_rn(1.0f, beta1), omb2 = __fsub_rn(1.0f, beta2);
    auto upd = [&](float gi, float& p, float& m, float& v) {
        const float g = __fmul_rn(gi, coef);
        p = __fmul_rn(p, decay);
        m = __fmaf_rn(beta1, m, __fmul_rn(omb1, g));
        v = __fmaf_rn(beta2, v, __fmul_rn(__fmul_rn(omb2, g), g));
        const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), bc2_sqrt), eps);
        p = __fmaf_rn(-step, __fdiv_rn(m, denom), p);
    };
    // 16-byte accesses, two quads per thread in flight (same per-element arithmetic as the scalar tail below)
    const bool al = ((reinterpret_cast<uintptr_t>(c.g) | reinterpret_cast<uintptr_t>(c.p) | reinterpret_cast<uintptr_t>(c.m) |
                      reinterpret_cast<uintptr_t>(c.v)) & 15) == 0;
    const int n4 = al ? (c.n >> 2) : 0;
    float4* p4 = reinterpret_cast<float4*>(c.p);
    float4* m4 = reinterpret_cast<float4*>(c.m);
    float4* v4 = reinterpret_cast<float4*>(c.v);
    const float4* g4 = reinterpret_cast<const float4*>(c.g);
    for (int i = threadIdx.x; i < n4; i += 512) {
        const int j = i + 256;
        const bool two = j < n4;
        const float4 ga = g4[i], gb = two ? g4[j] : ga;
        float4 pa = p4[i], ma = m4[i], va = v4[i];
        float4 pb = two ? p4[j] : pa, mb = two ? m4[j] : ma, vb = two ? v4[j] : va;
        upd(ga.x, pa.x, ma.x, va.x); upd(ga.y, pa.y, ma.y, va.y); upd(ga.z, pa.z, ma.z, va.z); upd(ga.w, pa.w, ma.w, va.w);
        upd(gb.x, pb.x, mb.x, vb.x); upd(gb.y, pb.y, mb.y, vb.y); upd(gb.z, pb.z, mb.z, vb.z); upd(gb.w, pb.w, mb.w, vb.w);
        p4[i] = pa; m4[i] = ma; v4[i] = va;
        if (two) { p4[j] = pb; m4[j] = mb; v4[j] = vb; }
    }
    for (int i = 4 * n4 + threadIdx.x; i < c.n; i += 256) {
        float p = c.p[i], m = c.m[i], v = c.v[i];
        upd(c.g[i], p, m, v);
        c.p[i] = p; c.m[i] = m; c.v[i] = v;
    }
}

extern "C" int mvit_grad_norm(const void* chunk_table, int nchunks, float max_norm, float* partials, float* out2,
                              void* stream) {
    if (!chunk_table || !partials || !out2 || nchunks <= 0) return MVIT_EINVAL;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(sqnorm_chunks_kernel, dim3(nchunks), dim3(256), 0, st, (const MtChunk*)chunk_table, partials);
    MVIT_LAUNCH_CHECK();
    hipLaunchKernelGGL(gradnorm_finish_kernel, dim3(1), dim3(64), 0, st, partials, nchunks, max_norm, out2);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}

extern "C" int mvit_adamw_step(const void* chunk_table, int nchunks, const float* norm_coef, float lr, float beta1,
                               float beta2, float eps, int step, void* stream) {
    if (!chunk_table || nchunks <= 0 || step <= 0) return MVIT_EINVAL;
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adamw_chunks_kernel, dim3(nchunks), dim3(256), 0, as_stream(stream), (const MtChunk*)chunk_table, norm_coef,
                       lr, beta1, beta2, eps, bc1, bc2s, (const float*)nullptr);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}

// The same step with its per-iteration scalars {lr, 1 - beta1^t, sqrt(1 - beta2^t)} read from device memory (3 floats): the
// launch carries no value that changes from step to step, so it can sit in a captured hipGraph that is replayed every iteration.
extern "C" int mvit_adamw_step_dev(const void* chunk_table, int nchunks, const float* norm_coef, const float* hyper, float beta1,
                                   float beta2, float eps, void* stream) {
    if (!chunk_table || nchunks <= 0 || !hyper) return MVIT_EINVAL;
    hipLaunchKernelGGL(adamw_chunks_kernel, dim3(nchunks), dim3(256), 0, as_stream(stream), (const MtChunk*)chunk_table, norm_coef,
                       0.f, beta1, beta2, eps, 1.f, 1.f, hyper);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}

extern "C" int mvit_mt_chunk_bytes(void) { return (int)sizeof(MtChunk); }
