"""cpu_nms / cpu_soft_nms -- replace the reference's Cython module utils/nms/cpu_nms.pyx
with the C++ host functions of libctdet (`ct_cpu_nms`, `ct_cpu_soft_nms`)."""
import numpy as np

from ctdet import ops


def cpu_nms(dets, thresh):
    """utils/nms/cpu_nms.pyx:17-68: greedy NMS, suppress IoU >= thresh, +1 convention."""
    return [int(i) for i in ops.cpu_nms(np.asarray(dets, dtype=np.float32), thresh, ge=True)]


def cpu_soft_nms(boxes, sigma=0.5, Nt=0.3, threshold=0.001, method=0):
    """utils/nms/cpu_nms.pyx:70-163: in-place soft-NMS, returns list(range(N'))."""
    return list(range(ops.cpu_soft_nms(boxes, sigma, Nt, threshold, method)))
