"""ORACLE (test infrastructure only): the test-time input transform.

Restates data/data_augment.py:224-266 (BaseTransform.__call__): cv2.resize(img, (S,S),
INTER_LINEAR) -> float32 -> minus means -> transpose(2,0,1).  cv2 is a third-party dependency
that is absent from /root/reference and from this image (opencv-python, unpinned in the
reference's README); the resize below restates OpenCV's published 8-bit bilinear algorithm
(modules/imgproc/src/resize.cpp: resizeGeneric_ / HResizeLinear / VResizeLinear with
INTER_RESIZE_COEF_BITS = 11).  PARITY UNPINNED against cv2 itself: the tests pin it to
known answers and to float bilinear interpolation (torch, align_corners=False) within 1 grey level.
"""
import numpy as np


def _taps(n_dst, n_src, zero_frac_at_edges):
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * (float(n_src) / n_dst) - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    if zero_frac_at_edges:
        lo = s < 0
        f[lo], s[lo] = 0, 0
        hi = s >= n_src - 1
        f[hi], s[hi] = 0, n_src - 1
        s0, s1 = s, np.minimum(s + 1, n_src - 1)
    else:
        s0, s1 = np.clip(s, 0, n_src - 1), np.clip(s + 1, 0, n_src - 1)
    c0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
    c1 = np.rint(f * np.float32(2048)).astype(np.int64)
    return s0, s1, c0, c1


def resize_linear_u8(img, size):
    """uint8 [H,W,C] -> uint8 [size,size,C]."""
    img = np.asarray(img)
    H, W = img.shape[:2]
    x0, x1, a0, a1 = _taps(size, W, True)
    y0, y1, b0, b1 = _taps(size, H, False)
    src = img.astype(np.int64)
    hor = src[:, x0, :] * a0[None, :, None] + src[:, x1, :] * a1[None, :, None]      # [H,size,C]
    r0, r1 = hor[y0], hor[y1]
    v = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def base_transform(img, size, means):
    """-> float32 [C,size,size]."""
    out = resize_linear_u8(img, size).astype(np.float32)
    out -= np.asarray(means, dtype=np.float32)
    return out.transpose(2, 0, 1).copy()
