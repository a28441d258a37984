"""NMS dispatch -- drop-in for the reference's utils/nms_wrapper.py:23-31."""
from .nms.cpu_nms import cpu_nms, cpu_soft_nms
from .nms.gpu_nms import gpu_nms


def nms(dets, thresh, force_cpu=False):
    """dets: float32 [n,5] host array.  Returns indices usable as `dets[keep, :]`, in
    descending-score order.  force_cpu -> the reference's CPU rule (IoU >= thresh), otherwise
    the device kernel with the CUDA rule (IoU > thresh)."""
    if dets.shape[0] == 0:
        return []
    if force_cpu:
        return cpu_nms(dets, thresh)
    return gpu_nms(dets, thresh)
