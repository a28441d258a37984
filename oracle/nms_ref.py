"""Oracle (test infrastructure, NOT product): greedy NMS with the +1 pixel convention.

Restates
  utils/nms/py_cpu_nms.py:10-38      py_cpu_nms   (suppress IoU >  thresh; same rule as
  utils/nms/nms_kernel.cu:24-32,71   devIoU / nms_kernel, the CUDA path `gpu_nms`)
  utils/nms/cpu_nms.pyx:17-68        cpu_nms      (suppress IoU >= thresh)
  utils/nms/cpu_nms.pyx:70-163       cpu_soft_nms (in-place soft-NMS)
  utils/nms/gpu_nms.pyx:16-31        argsort-desc -> _nms -> order[keep]
  test.py:136-161                    per-class select / NMS / top-200 per image

fp32 expression order is the reference's: areas = (x2-x1+1)*(y2-y1+1);
w = max(0, xx2-xx1+1); inter = w*h; ovr = inter / (area_i + area_j - inter).

Tie order: the reference uses numpy's unstable ``scores.argsort()[::-1]`` whose tie
order is implementation-defined.  The build DEFINES the order as
"descending score, lower original index first" (``stable_desc_order``); goldens are
tie-free so both agree.

The compiled C twin (oracle/nms_ref.c -> oracle/_build/libnms_ref.so) is used for the
full-size cases and the CPU baseline; it is checked against this file in tests.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def stable_desc_order(scores):
    """Descending score, ties broken by lower index first (the build's defined order)."""
    return np.argsort(-scores.astype(np.float32), kind='stable')


def nms_sorted(dets_sorted, thresh, ge=False):
    """Greedy NMS on boxes ALREADY sorted by descending score (the `_nms` contract,
    utils/nms/nms_kernel.cu:91-144).  Returns ascending indices into the sorted array."""
    d = np.ascontiguousarray(dets_sorted, dtype=np.float32)
    n = d.shape[0]
    x1, y1, x2, y2 = d[:, 0], d[:, 1], d[:, 2], d[:, 3]
    one = np.float32(1)
    areas = (x2 - x1 + one) * (y2 - y1 + one)
    t = np.float32(thresh)
    alive = np.ones(n, dtype=bool)
    keep = []
    for i in range(n):
        if not alive[i]:
            continue
        keep.append(i)
        if i + 1 >= n:
            break
        xx1 = np.maximum(x1[i], x1[i + 1:])
        yy1 = np.maximum(y1[i], y1[i + 1:])
        xx2 = np.minimum(x2[i], x2[i + 1:])
        yy2 = np.minimum(y2[i], y2[i + 1:])
        w = np.maximum(np.float32(0), xx2 - xx1 + one)
        h = np.maximum(np.float32(0), yy2 - yy1 + one)
        inter = w * h
        with np.errstate(divide='ignore', invalid='ignore'):
            ovr = inter / (areas[i] + areas[i + 1:] - inter)
        sup = (ovr >= t) if ge else (ovr > t)
        alive[i + 1:] &= ~sup
    return np.asarray(keep, dtype=np.int64)


def nms(dets, thresh, ge=False, order=None):
    """py_cpu_nms / gpu_nms (ge=False) or cpu_nms (ge=True) on unsorted dets [n,5].
    Returns original indices in descending-score order (what `c_dets[keep]` expects)."""
    dets = np.asarray(dets, dtype=np.float32)
    if dets.shape[0] == 0:
        return np.zeros(0, dtype=np.int64)
    if order is None:
        order = stable_desc_order(dets[:, 4])
    keep_sorted = nms_sorted(dets[order], thresh, ge=ge)
    return order[keep_sorted]


def soft_nms(boxes, sigma=0.5, Nt=0.3, threshold=0.001, method=0):
    """utils/nms/cpu_nms.pyx:70-163.  Mutates `boxes` ([N,5] float32) in place; returns N'.
    Arithmetic is C float (fp32) except `np.exp(-(ov*ov)/sigma)` which the reference
    evaluates in double before the fp32 store of weight."""
    b = boxes
    f = np.float32
    N = b.shape[0]
    i = 0
    while i < N:
        maxscore = b[i, 4]
        maxpos = i
        t = b[i].copy()
        pos = i + 1
        while pos < N:
            if maxscore < b[pos, 4]:
                maxscore = b[pos, 4]
                maxpos = pos
            pos += 1
        b[i] = b[maxpos]
        b[maxpos] = t
        tx1, ty1, tx2, ty2 = b[i, 0], b[i, 1], b[i, 2], b[i, 3]
        pos = i + 1
        while pos < N:
            x1, y1, x2, y2 = b[pos, 0], b[pos, 1], b[pos, 2], b[pos, 3]
            area = f(f(x2 - x1 + f(1)) * f(y2 - y1 + f(1)))
            iw = f(min(tx2, x2) - max(tx1, x1) + f(1))
            if iw > 0:
                ih = f(min(ty2, y2) - max(ty1, y1) + f(1))
                if ih > 0:
                    ua = f(f(f(tx2 - tx1 + f(1)) * f(ty2 - ty1 + f(1))) + area - f(iw * ih))
                    ov = f(f(iw * ih) / ua)
                    if method == 1:
                        weight = f(1) - ov if ov > f(Nt) else f(1)
                    elif method == 2:
                        weight = f(np.exp(-(np.float64(ov) * np.float64(ov)) / np.float64(f(sigma))))
                    else:
                        weight = f(0) if ov > f(Nt) else f(1)
                    b[pos, 4] = f(weight * b[pos, 4])
                    if b[pos, 4] < f(threshold):
                        b[pos] = b[N - 1]
                        N -= 1
                        pos -= 1
            pos += 1
        i += 1
    return N


# ---------------------------------------------------------------------------
# test.py:136-161 -- per-image post-processing after Detect
# ---------------------------------------------------------------------------
def postprocess_image(boxes, scores, scale_wh, conf_thresh=0.01, nms_thresh=0.45,
                      max_per_image=200, ge=False, nms_fn=None):
    """boxes [P,4] normalised, scores [P,1+T] -> list over classes 1..T of [k,5] float32.

    `boxes *= scale` in fp32 (test.py:136), `score > thresh` (:143), NMS 0.45 (:152),
    global top `max_per_image` by score threshold `>=` k-th largest (:155-161)."""
    w, h = scale_wh
    scale = np.array([w, h, w, h], dtype=np.float32)
    boxes = (np.asarray(boxes, dtype=np.float32) * scale).astype(np.float32)
    scores = np.asarray(scores, dtype=np.float32)
    ncls = scores.shape[1]
    out = [np.empty((0, 5), dtype=np.float32)]  # class 0 placeholder
    fn = nms_fn or (lambda d, t: nms(d, t, ge=ge))
    for j in range(1, ncls):
        inds = np.where(scores[:, j] > np.float32(conf_thresh))[0]
        if len(inds) == 0:
            out.append(np.empty((0, 5), dtype=np.float32))
            continue
        c_dets = np.hstack((boxes[inds], scores[inds, j][:, None])).astype(np.float32, copy=False)
        keep = fn(c_dets, nms_thresh)
        out.append(c_dets[keep, :])
    if max_per_image > 0:
        image_scores = np.hstack([out[j][:, -1] for j in range(1, ncls)])
        if len(image_scores) > max_per_image:
            image_thresh = np.sort(image_scores)[-max_per_image]
            for j in range(1, ncls):
                k = np.where(out[j][:, -1] >= image_thresh)[0]
                out[j] = out[j][k, :]
    return out


# ---------------------------------------------------------------------------
# C twin (fast path for full-size problems and the CPU baseline)
# ---------------------------------------------------------------------------
_clib = None


def build_c(force=False):
    """Compile oracle/nms_ref.c -> oracle/_build/libnms_ref.so (gcc, -O2, no fast-math, no FMA)."""
    out_dir = os.path.join(_HERE, '_build')
    so = os.path.join(out_dir, 'libnms_ref.so')
    src = os.path.join(_HERE, 'nms_ref.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        os.makedirs(out_dir, exist_ok=True)
        subprocess.check_call(['gcc', '-O2', '-fPIC', '-shared', '-ffp-contract=off', '-std=c99',
                               '-o', so, src, '-lm'])
    return so


def _c():
    global _clib
    if _clib is None:
        lib = ctypes.CDLL(build_c())
        lib.oracle_nms_sorted.restype = ctypes.c_int
        lib.oracle_nms_sorted.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_float,
                                          ctypes.c_int, ctypes.c_void_p]
        _clib = lib
    return _clib


def nms_sorted_c(dets_sorted, thresh, ge=False):
    d = np.ascontiguousarray(dets_sorted, dtype=np.float32)
    n = d.shape[0]
    keep = np.empty(max(n, 1), dtype=np.int32)
    k = _c().oracle_nms_sorted(d.ctypes.data, n, ctypes.c_float(thresh), int(bool(ge)), keep.ctypes.data)
    return keep[:k].astype(np.int64)


def nms_c(dets, thresh, ge=False, order=None):
    dets = np.asarray(dets, dtype=np.float32)
    if dets.shape[0] == 0:
        return np.zeros(0, dtype=np.int64)
    if order is None:
        order = stable_desc_order(dets[:, 4])
    return order[nms_sorted_c(dets[order], thresh, ge=ge)]
