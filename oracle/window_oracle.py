"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rules as mvit_oracle.py).

CPU restatement of the sliding-window front end of scripts/run_action_classification_temporal_inf.py:
window list, per-window frame indices, ROI resize to frame_size x frame_size, /255 and mean/std normalisation.

Parity status:
  * proposals / frame indices: PINNED (pure integer logic of scripts/module_wrapper.py:246-253,384-397, checked against the
    torch.linspace formula the reference calls).
  * resize: the reference calls cv2.resize(uint8 image, (S, S), interpolation=cv2.INTER_LINEAR) (scripts/utils.py:172-211,
    keep_scale=False).  cv2 is NOT installed in this image and is not vendored by the reference, so this is a restatement
    of OpenCV's published 8-bit bilinear algorithm (opencv/modules/imgproc/src/resize.cpp, 4.x: 11-bit fixed-point
    coefficients, horizontal pass in int, vertical pass ((b0*(S0>>4))>>16)+((b1*(S1>>4))>>16)+2)>>2).  PARITY UNPINNED
    for this step: no golden vector from a real cv2 could be produced here.
  * exact 2x decimation (an 896 x 896 ROI -> 448): cv2.resize switches INTER_LINEAR to INTER_AREA there (resize.cpp, resize():
    ``if (interpolation == INTER_LINEAR && is_area_fast && iscale_x == 2 && iscale_y == 2) interpolation = INTER_AREA``, with the
    comment that the two are equal).  They ARE equal in the 8-bit path, so no second code path is needed: at 2x the bilinear
    coefficients are 1024 / 1024 exactly ((d + .5) * 2 - .5 = 2d + .5), the horizontal pass gives S = 1024 (a + b), and
    ((1024 * (S >> 4)) >> 16) = a + b without any truncation, so the result is (a + b + c + d + 2) >> 2 -- which is what the
    INTER_AREA fast path computes for uchar (ResizeAreaFastVec: (S[0] + S[1] + nextS[0] + nextS[1] + 2) >> 2).
    ``resize_area_fast_2x_u8`` below restates that path; tests/test_inference_cpu.py and tests/test_hip_inference.py assert the
    equality for the restatement and for the HIP kernel.  (Other integer decimation factors do NOT switch: only 2 x 2.)
"""
import numpy as np
import torch

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def get_proposals(num_frames, prop_length=64, prop_stride=16):
    """scripts/module_wrapper.py:246-253 -- windows run past the end of the video on purpose."""
    return [(i, i + prop_length) for i in range(0, num_frames, prop_stride)]


def frame_idxs_uniform(t0, t1, frame_length, video_num_frame):
    """scripts/module_wrapper.py:384-397."""
    index = torch.linspace(t0, t1, frame_length)
    return torch.clamp(index, 0, video_num_frame - 1).long().numpy().tolist()


def _coeffs(src, dst):
    scale = 1.0 / (float(dst) / float(src))          # double, as resize.cpp computes scale_x from inv_scale_x
    ofs = np.zeros(dst, np.int32)
    alpha = np.zeros((dst, 2), np.int32)
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - s)
        if s < 0:
            s, f = 0, np.float32(0)
        if s >= src - 1:
            s, f = src - 1, np.float32(0)
        ofs[d] = s
        a0 = np.float32(1.0) - f
        alpha[d, 0] = int(np.rint(np.float32(a0 * COEF_SCALE)))
        alpha[d, 1] = int(np.rint(np.float32(f * COEF_SCALE)))
    return ofs, alpha


def resize_linear_u8(img, dst_h, dst_w):
    """img uint8 [H,W,C] -> uint8 [dst_h,dst_w,C], OpenCV INTER_LINEAR 8-bit path."""
    H, W, C = img.shape
    xofs, xa = _coeffs(W, dst_w)
    yofs, ya = _coeffs(H, dst_h)
    src = img.astype(np.int32)
    x1 = np.minimum(xofs + 1, W - 1)
    hres = src[:, xofs, :] * xa[None, :, 0, None] + src[:, x1, :] * xa[None, :, 1, None]      # [H, dst_w, C] int32
    y1 = np.minimum(yofs + 1, H - 1)
    S0 = hres[yofs]
    S1 = hres[y1]
    b0 = ya[:, 0][:, None, None]
    b1 = ya[:, 1][:, None, None]
    out = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def resize_area_fast_2x_u8(img):
    """OpenCV INTER_AREA fast path for an exact 2 x 2 decimation of uint8 (resize.cpp, ResizeAreaFast_Invoker / ResizeAreaFastVec,
    scale 2): out = (a + b + c + d + 2) >> 2 over each 2 x 2 cell -- the path cv2.resize(..., INTER_LINEAR) takes at this scale."""
    H, W, C = img.shape
    assert H % 2 == 0 and W % 2 == 0
    s = img.astype(np.int32)
    return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)


def resize_bilinear_float(img, dst_h, dst_w):
    """Independent cross-check (NOT OpenCV's arithmetic): bilinear interpolation with half-pixel centres in float64, edge-clamped,
    rounded to nearest.  OpenCV's 11-bit fixed-point result stays within 1 LSB of it."""
    H, W, C = img.shape

    def axis(src, dst):
        f = (np.arange(dst) + 0.5) * (src / dst) - 0.5
        s = np.floor(f).astype(np.int64)
        w = f - s
        w = np.where(s < 0, 0.0, w)
        s = np.maximum(s, 0)
        w = np.where(s >= src - 1, 0.0, w)
        s = np.minimum(s, src - 1)
        return s, np.minimum(s + 1, src - 1), w
    y0, y1, wy = axis(H, dst_h)
    x0, x1, wx = axis(W, dst_w)
    a = img.astype(np.float64)
    top = a[y0][:, x0] * (1 - wx)[None, :, None] + a[y0][:, x1] * wx[None, :, None]
    bot = a[y1][:, x0] * (1 - wx)[None, :, None] + a[y1][:, x1] * wx[None, :, None]
    return top * (1 - wy)[:, None, None] + bot * wy[:, None, None]


def preprocess_window(frames_u8, idxs, size, mean=0.45, std=0.225):
    """frames_u8 [N,H,W,3] uint8, idxs list of 16 frame indices -> float32 [3,16,size,size]
    (scripts/module_wrapper.py:304-341: resize, /255, [T,H,W,C]->[C,T,H,W], (x-mean)/std)."""
    out = np.stack([resize_linear_u8(frames_u8[i], size, size) for i in idxs]).astype(np.float32)
    out /= 255.0
    out = out.transpose(3, 0, 1, 2)
    return ((out - np.float32(mean)) / np.float32(std)).astype(np.float32)
