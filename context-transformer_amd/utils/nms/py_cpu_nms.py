"""py_cpu_nms -- the 'pure Python NMS baseline' of the reference (utils/nms/py_cpu_nms.py:10-38),
kept for API completeness: numpy, suppress IoU > thresh, +1 convention."""
import numpy as np


def py_cpu_nms(dets, thresh):
    d = np.asarray(dets)
    area = (d[:, 2] - d[:, 0] + 1) * (d[:, 3] - d[:, 1] + 1)
    todo = np.argsort(-d[:, 4], kind='stable')
    keep = []
    while todo.size:
        top, rest = todo[0], todo[1:]
        keep.append(top)
        w = np.maximum(0.0, np.minimum(d[top, 2], d[rest, 2]) - np.maximum(d[top, 0], d[rest, 0]) + 1)
        h = np.maximum(0.0, np.minimum(d[top, 3], d[rest, 3]) - np.maximum(d[top, 1], d[rest, 1]) + 1)
        inter = w * h
        todo = rest[inter / (area[top] + area[rest] - inter) <= thresh]
    return keep
