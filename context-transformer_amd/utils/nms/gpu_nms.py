"""gpu_nms -- replaces the reference's Cython shim utils/nms/gpu_nms.pyx:16-31.

argsort by descending score on the host (ties: lower index first -- the reference's tie order
is numpy-implementation-defined), then the `_nms`-contract entry point of libctdet
(`ct_nms_sorted_host`, the HIP twin of utils/nms/nms_kernel.cu:91-144), then map back.
"""
import numpy as np

from ctdet import ops


def gpu_nms(dets, thresh, device_id=0):
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    if dets.shape[0] == 0:
        return []
    order = np.argsort(-dets[:, 4], kind='stable')
    keep = ops.nms_sorted_host(dets[order, :], thresh, ge=False, device_id=device_id)
    return list(order[keep])
