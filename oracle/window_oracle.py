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


def preprocess_window(frames_u8, idxs, size, mean=0.45, std=0.225):
    """frames_u8 [N,H,W,3] uint8, idxs list of 16 frame indices -> float32 [3,16,size,size]
    (scripts/module_wrapper.py:304-341: resize, /255, [T,H,W,C]->[C,T,H,W], (x-mean)/std)."""
    out = np.stack([resize_linear_u8(frames_u8[i], size, size) for i in idxs]).astype(np.float32)
    out /= 255.0
    out = out.transpose(3, 0, 1, 2)
    return ((out - np.float32(mean)) / np.float32(std)).astype(np.float32)
