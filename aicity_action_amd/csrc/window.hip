// Sliding-window front end on the GPU: for every window, gather its 16 frames from the
// decoded uint8 stream [N][H][W][3] by index, resize to SxS with OpenCV's 8-bit INTER_LINEAR arithmetic (11-bit fixed
// point coefficients; restated from opencv/modules/imgproc/src/resize.cpp because the reference calls cv2.resize,
// scripts/utils.py:172-211), then /255, (x-mean)/std and write [B][3][16][S][S] fp32 -- the exact tensor MViT.forward
// consumes.  One thread per output pixel (all 3 channels); byte traffic: 4 source pixels x 3 B in, 12 B out: HBM-bound.
#include "common.h"

__device__ __forceinline__ void lin_coef(int d, int src, double scale, int& s, int& a0, int& a1) {
    float f = (float)((d + 0.5) * scale - 0.5);
    s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { s = 0; f = 0.f; }
    if (s >= src - 1) { s = src - 1; f = 0.f; }
    a0 = (int)rintf((1.0f - f) * 2048.0f);
    a1 = (int)rintf(f * 2048.0f);
}

__global__ __launch_bounds__(256) void window_preprocess_kernel(const uint8_t* __restrict__ frames, const int* __restrict__ idx,
                                                                float* __restrict__ out, int H, int W, int S, int nclips,
                                                                int fl, double sx, double sy, float mean, float stdv) {
    const int64_t total = (int64_t)nclips * fl * S * S;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % S);
        int64_t r = i / S;
        const int y = (int)(r % S); r /= S;
        const int t = (int)(r % fl);
        const int b = (int)(r / fl);
        int xs, xa0, xa1, ys, ya0, ya1;
        lin_coef(x, W, sx, xs, xa0, xa1);
        lin_coef(y, H, sy, ys, ya0, ya1);
        const int x1 = xs + 1 < W ? xs + 1 : W - 1, y1 = ys + 1 < H ? ys + 1 : H - 1;
        const uint8_t* f = frames + (int64_t)idx[b * fl + t] * H * W * 3;
        const uint8_t* p00 = f + ((int64_t)ys * W + xs) * 3;
        const uint8_t* p01 = f + ((int64_t)ys * W + x1) * 3;
        const uint8_t* p10 = f + ((int64_t)y1 * W + xs) * 3;
        const uint8_t* p11 = f + ((int64_t)y1 * W + x1) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int S0 = (int)p00[c] * xa0 + (int)p01[c] * xa1;
            const int S1 = (int)p10[c] * xa0 + (int)p11[c] * xa1;
            int v = (((ya0 * (S0 >> 4)) >> 16) + ((ya1 * (S1 >> 4)) >> 16) + 2) >> 2;
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            const float fv = ((float)v / 255.0f - mean) / stdv;
            out[((((int64_t)b * 3 + c) * fl + t) * S + y) * S + x] = fv;
        }
    }
}

// frames: device uint8 [N][H][W][3]; frame_idx: device int32 [nclips][frame_length]; out fp32 [nclips][3][frame_length][S][S]
extern "C" int mvit_window_preprocess(const void* frames, const int* frame_idx, float* out, int H, int W, int S, int nclips,
                                      int frame_length, float mean, float std, void* stream) {
    if (!frames || !frame_idx || !out || H <= 0 || W <= 0 || S <= 0 || nclips <= 0 || frame_length <= 0 || std == 0.f)
        return MVIT_EINVAL;
    const int64_t total = (int64_t)nclips * frame_length * S * S;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(window_preprocess_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (const uint8_t*)frames,
                       frame_idx, out, H, W, S, nclips, frame_length, 1.0 / ((double)S / (double)W), 1.0 / ((double)S / (double)H),
                       mean, std);
    MVIT_LAUNCH_CHECK();
    return MVIT_OK;
}
