"""ORACLE (test infrastructure only): pixels of the training-time augmentation.

Restates data/data_augment.py:82-161 and :164-221 for GIVEN random decisions (a plan dict as produced by the product's
`preproc.decide`): crop (:57) -> `_distort` (:82-110) -> `_expand` (:113-146) -> `_mirror` (:149-155) ->
`preproc_for_test` (:158-165: cv2.resize, float32, minus mean, CHW), materialising every intermediate image the way
the reference does.  cv2 is absent from /root/reference and from this image; `cvtColor` (8-bit BGR<->HSV, H in
[0,180)) and `resize` restate OpenCV's published formulas in float64 with round-to-nearest 8-bit results.
PARITY UNPINNED against cv2 itself (its integer tables can differ from this by one grey level); the decision
logic and the box arithmetic ARE pinned by the reference (tests/golden/augment.npz).
"""
import numpy as np


def _convert(img, alpha=1.0, beta=0.0):
    tmp = img.astype(float) * alpha + beta
    tmp[tmp < 0] = 0
    tmp[tmp > 255] = 255
    return tmp.astype(np.uint8)                 # `image[:] = tmp` on a uint8 array truncates


def bgr2hsv_u8(img):
    """8-bit cv2.cvtColor(BGR2HSV): V = max, S = 255*(V-min)/V, H = 30 * sector position, both rounded half up in
    exact integer arithmetic (OpenCV's sdiv/hdiv fixed-point tables approximate exactly that)."""
    b, g, r = [img[..., i].astype(np.int64) for i in range(3)]
    v = np.maximum(b, np.maximum(g, r))
    mn = np.minimum(b, np.minimum(g, r))
    diff = v - mn
    s = np.where(v == 0, 0, (2 * 255 * diff + v) // np.where(v == 0, 1, 2 * v))
    num = np.where(v == r, g - b, np.where(v == g, 2 * diff + (b - r), 4 * diff + (r - g)))
    num = np.where(num < 0, num + 6 * diff, num)
    h = np.where(diff == 0, 0, (60 * num + diff) // np.where(diff == 0, 1, 2 * diff))
    h = np.where(h >= 180, h - 180, h)
    return np.stack([h, s, v], -1).astype(np.uint8)


def hsv2bgr_u8(img):
    h, s, v = [img[..., i].astype(np.float64) for i in range(3)]
    S = s / 255.0
    hh = h / 30.0
    sec = np.minimum(hh.astype(np.int64), 5)
    f = hh - sec
    p, q, t = v * (1 - S), v * (1 - S * f), v * (1 - S * (1 - f))
    R = np.choose(sec, [v, q, p, p, t, v])
    G = np.choose(sec, [t, v, v, q, p, p])
    B = np.choose(sec, [p, p, t, v, v, q])
    return np.clip(np.rint(np.stack([B, G, R], -1)), 0, 255).astype(np.uint8)


def distort(img, plan):
    img = img.copy()
    if plan['flags'] & 1:
        img = _convert(img, beta=np.float32(plan['beta']))
    if plan['flags'] & 2:
        img = _convert(img, alpha=np.float32(plan['alpha']))
    hsv = bgr2hsv_u8(img)
    if plan['flags'] & 4:
        hsv[..., 0] = ((hsv[..., 0].astype(int) + plan['hue']) % 180).astype(np.uint8)
    if plan['flags'] & 8:
        hsv[..., 1] = _convert(hsv[..., 1], alpha=np.float32(plan['sat']))
    return hsv2bgr_u8(hsv)


def resize_u8(img, S, interp):
    """float64 restatement of cv2.resize(img, (S, S)) for INTER_LINEAR (0), INTER_NEAREST (1), INTER_AREA (2)."""
    H, W = img.shape[:2]
    src = img.astype(np.float64)
    sx, sy = np.float32(W) / np.float32(S), np.float32(H) / np.float32(S)
    if interp == 1:
        xs = np.minimum(np.floor(np.arange(S, dtype=np.float32) * sx).astype(int), W - 1)
        ys = np.minimum(np.floor(np.arange(S, dtype=np.float32) * sy).astype(int), H - 1)
        return img[ys][:, xs]
    if interp == 2 and (sx > 1 or sy > 1):
        out = np.zeros((S, S, 3))
        def weights(n_dst, n_src, sc):
            w = np.zeros((n_dst, n_src))
            for d in range(n_dst):
                a, b = np.float32(d) * sc, min(np.float32(d + 1) * sc, np.float32(n_src))
                for i in range(int(np.floor(a)), int(np.ceil(b))):
                    w[d, i] = min(b, i + 1.0) - max(a, float(i))
            return w
        wy, wx = weights(S, H, sy), weights(S, W, sx)
        for c in range(3):
            out[..., c] = (wy @ src[..., c] @ wx.T) / (wy.sum(1)[:, None] * wx.sum(1)[None, :])
        return np.clip(np.rint(out), 0, 255).astype(np.uint8)
    fx = (np.arange(S, dtype=np.float32) + np.float32(0.5)) * sx - np.float32(0.5)
    fy = (np.arange(S, dtype=np.float32) + np.float32(0.5)) * sy - np.float32(0.5)
    ix, iy = np.floor(fx).astype(int), np.floor(fy).astype(int)
    ax, ay = (fx - ix).astype(np.float64), (fy - iy).astype(np.float64)
    lo, hi = ix < 0, ix >= W - 1
    ax[lo], ix[lo] = 0, 0
    ax[hi], ix[hi] = 0, W - 1
    ix1 = np.minimum(ix + 1, W - 1)
    iy0, iy1 = np.clip(iy, 0, H - 1), np.clip(iy + 1, 0, H - 1)
    top = src[iy0][:, ix] * (1 - ax)[None, :, None] + src[iy0][:, ix1] * ax[None, :, None]
    bot = src[iy1][:, ix] * (1 - ax)[None, :, None] + src[iy1][:, ix1] * ax[None, :, None]
    out = top * (1 - ay)[:, None, None] + bot * ay[:, None, None]
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def augment(img, plan, size, means):
    """uint8 [H,W,3] + plan -> float32 [3,size,size] (what `preproc.__call__` returns as its image)."""
    l, t, w, h = plan['crop']
    x = img[t:t + h, l:l + w]
    x = distort(x, plan) if plan['flags'] else x.copy()
    ew, eh, left, top = plan['exp']
    if (ew, eh) != (w, h):
        canvas = np.empty((eh, ew, 3), dtype=np.uint8)
        canvas[:, :] = np.asarray(means)                 # float means cast into the uint8 canvas
        canvas[top:top + h, left:left + w] = x
        x = canvas
    if plan['mirror']:
        x = x[:, ::-1]
    out = resize_u8(x, size, plan['interp']).astype(np.float32)
    out -= np.asarray(means, dtype=np.float32)
    return out.transpose(2, 0, 1).copy()
