"""NMS dispatch -- the entry point the reference's test.py imports (`from utils.nms_wrapper import nms`,
reference utils/nms_wrapper.py:23-31).  Two suppression rules live behind it, exactly as in the reference:

    force_cpu=False   device kernel, a box is suppressed when IoU >  thresh   (utils/nms/nms_kernel.cu:71)
    force_cpu=True    host code,     a box is suppressed when IoU >= thresh   (utils/nms/cpu_nms.pyx:65)

Both return indices into `dets` in descending-score order, usable as `dets[keep, :]`.
"""
import numpy as np

from .nms import cpu_nms as _host
from .nms import gpu_nms as _device

cpu_nms, cpu_soft_nms, gpu_nms = _host.cpu_nms, _host.cpu_soft_nms, _device.gpu_nms
_RULES = {False: gpu_nms, True: cpu_nms}


def nms(dets, thresh, force_cpu=False):
    """dets: float32 [n, 5] host array of (x1, y1, x2, y2, score) rows; an empty input gives []."""
    if np.shape(dets)[0] == 0:
        return []
    return _RULES[bool(force_cpu)](dets, thresh)
