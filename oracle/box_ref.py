"""Oracle (test infrastructure, NOT product): prior boxes, box coding, IoU, matching.

Restates, op for op in torch-CPU fp32, the reference functions
  layers/functions/prior_box.py:31-56   PriorBox.forward
  utils/box_utils.py:5-14               point_form
  utils/box_utils.py:29-68              intersect / jaccard
  utils/box_utils.py:70-80              matrix_iou (numpy)
  utils/box_utils.py:83-132             match
  utils/box_utils.py:135-156            encode
  utils/box_utils.py:184-202            decode
  layers/functions/detection.py:18-55   Detect.forward
  utils/box_utils.py:238-302            nms (torch greedy, top_k)
so that results are bit-identical to the reference on CPU (same op order, same
fp32 roundings).  Pinned by tests/golden/box_ops.npz.
"""
import math

import numpy as np
import torch

# data/config.py:10-135 -- anchor configurations (data only).
_RFB_300_COMMON = dict(feature_maps=[38, 19, 10, 5, 3, 1], min_dim=300,
                       steps=[8, 16, 32, 64, 100, 300],
                       aspect_ratios=[[2, 3], [2, 3], [2, 3], [2, 3], [2], [2]],
                       variance=[0.1, 0.2], clip=True)
_RFB_512_COMMON = dict(feature_maps=[64, 32, 16, 8, 4, 2, 1], min_dim=512,
                       steps=[8, 16, 32, 64, 128, 256, 512],
                       aspect_ratios=[[2, 3], [2, 3], [2, 3], [2, 3], [2, 3], [2], [2]],
                       variance=[0.1, 0.2], clip=True)
ANCHOR_CFGS = {
    'VOC_300': dict(_RFB_300_COMMON, min_sizes=[30, 60, 111, 162, 213, 264],
                    max_sizes=[60, 111, 162, 213, 264, 315]),
    'COCO_300': dict(_RFB_300_COMMON, min_sizes=[21, 45, 99, 153, 207, 261],
                     max_sizes=[45, 99, 153, 207, 261, 315]),
    'VOC_512': dict(_RFB_512_COMMON,
                    min_sizes=[35.84, 76.8, 153.6, 230.4, 307.2, 384.0, 460.8],
                    max_sizes=[76.8, 153.6, 230.4, 307.2, 384.0, 460.8, 537.6]),
    'COCO_512': dict(_RFB_512_COMMON,
                     min_sizes=[20.48, 51.2, 133.12, 215.04, 296.96, 378.88, 460.8],
                     max_sizes=[51.2, 133.12, 215.04, 296.96, 378.88, 460.8, 542.72]),
    'VOC_SSD_300': dict(_RFB_300_COMMON, min_sizes=[30, 60, 111, 162, 213, 264],
                        max_sizes=[60, 111, 162, 213, 264, 315],
                        aspect_ratios=[[2], [2, 3], [2, 3], [2, 3], [2], [2]]),
    'COCO_SSD_300': dict(_RFB_300_COMMON, min_sizes=[21, 45, 99, 153, 207, 261],
                         max_sizes=[45, 99, 153, 207, 261, 315],
                         aspect_ratios=[[2], [2, 3], [2, 3], [2, 3], [2], [2]]),
    'COCO_mobile_300': dict(feature_maps=[19, 10, 5, 3, 2, 1], min_dim=300,
                            steps=[16, 32, 64, 100, 150, 300],
                            min_sizes=[45, 90, 135, 180, 225, 270],
                            max_sizes=[90, 135, 180, 225, 270, 315],
                            aspect_ratios=[[2, 3], [2, 3], [2, 3], [2, 3], [2], [2]],
                            variance=[0.1, 0.2], clip=True),
}


def prior_box(cfg):
    """layers/functions/prior_box.py:31-56: anchors (cx,cy,w,h), python doubles -> fp32, clamp."""
    rows = []
    size = cfg['min_dim']
    for k, f in enumerate(cfg['feature_maps']):
        f_k = size / cfg['steps'][k]
        s_k = cfg['min_sizes'][k] / size
        s_kp = math.sqrt(s_k * (cfg['max_sizes'][k] / size))
        for i in range(f):
            for j in range(f):
                cx = (j + 0.5) / f_k
                cy = (i + 0.5) / f_k
                rows.append((cx, cy, s_k, s_k))
                rows.append((cx, cy, s_kp, s_kp))
                for ar in cfg['aspect_ratios'][k]:
                    r = math.sqrt(ar)
                    rows.append((cx, cy, s_k * r, s_k / r))
                    rows.append((cx, cy, s_k / r, s_k * r))
    out = torch.tensor(rows, dtype=torch.float64).to(torch.float32)
    if cfg['clip']:
        out = out.clamp(min=0, max=1)
    return out


def point_form(boxes):
    """utils/box_utils.py:5-14."""
    return torch.cat((boxes[:, :2] - boxes[:, 2:] / 2, boxes[:, :2] + boxes[:, 2:] / 2), 1)


def jaccard(box_a, box_b):
    """utils/box_utils.py:29-68: pairwise IoU, no +1 convention; inter/(area_a+area_b-inter)."""
    ax1, ay1, ax2, ay2 = [box_a[:, i:i + 1] for i in range(4)]
    bx1, by1, bx2, by2 = [box_b[:, i].unsqueeze(0) for i in range(4)]
    w = torch.clamp(torch.min(ax2, bx2) - torch.max(ax1, bx1), min=0)
    h = torch.clamp(torch.min(ay2, by2) - torch.max(ay1, by1), min=0)
    inter = w * h
    area_a = (ax2 - ax1) * (ay2 - ay1)
    area_b = (bx2 - bx1) * (by2 - by1)
    union = area_a + area_b - inter
    return inter / union


def matrix_iou(a, b):
    """utils/box_utils.py:70-80 (numpy, used by augmentation only)."""
    lt = np.maximum(a[:, None, :2], b[:, :2])
    rb = np.minimum(a[:, None, 2:], b[:, 2:])
    area_i = np.prod(rb - lt, axis=2) * (lt < rb).all(axis=2)
    area_a = np.prod(a[:, 2:] - a[:, :2], axis=1)
    area_b = np.prod(b[:, 2:] - b[:, :2], axis=1)
    return area_i / (area_a[:, None] + area_b - area_i)


def encode(matched, priors, variances):
    """utils/box_utils.py:135-156."""
    g_cxcy = (matched[:, :2] + matched[:, 2:]) / 2 - priors[:, :2]
    g_cxcy = g_cxcy / (variances[0] * priors[:, 2:])
    g_wh = (matched[:, 2:] - matched[:, :2]) / priors[:, 2:]
    g_wh = torch.log(g_wh) / variances[1]
    return torch.cat([g_cxcy, g_wh], 1)


def decode(loc, priors, variances):
    """utils/box_utils.py:184-202.  NB x2y2 is computed from the already rounded x1y1."""
    cxcy = priors[:, :2] + loc[:, :2] * variances[0] * priors[:, 2:]
    wh = priors[:, 2:] * torch.exp(loc[:, 2:] * variances[1])
    x1y1 = cxcy - wh / 2
    x2y2 = wh + x1y1
    return torch.cat((x1y1, x2y2), 1)


def match(threshold, truths, priors, variances, labels):
    """utils/box_utils.py:83-132 for one image.

    truths [G,4] point form; labels [G,2] = (label, mixup weight).
    Returns (loc [P,4], conf [P,2], obj [P] bool, best_truth_overlap [P] before force-match).
    """
    overlaps = jaccard(truths, point_form(priors))
    best_prior_overlap, best_prior_idx = overlaps.max(1)
    best_truth_overlap, best_truth_idx = overlaps.max(0)
    raw_overlap = best_truth_overlap.clone()
    best_truth_overlap = best_truth_overlap.clone()
    best_truth_idx = best_truth_idx.clone()
    best_truth_overlap[best_prior_idx] = 2
    for j in range(best_prior_idx.numel()):      # :122-123 -- later GT wins on collisions
        best_truth_idx[best_prior_idx[j]] = j
    matches = truths[best_truth_idx]
    conf = labels[best_truth_idx].clone()
    low = best_truth_overlap < threshold
    conf[low, 0] = 0
    conf[low, 1] = 1
    loc = encode(matches, priors, variances)
    obj = conf[:, 0] != 0
    return loc, conf, obj, raw_overlap


def detect(loc, conf, obj, priors, variances=(0.1, 0.2)):
    """layers/functions/detection.py:18-55: per-image decode + scores=[obj0, obj1*conf]."""
    num = loc.shape[0]
    boxes = torch.zeros(num, priors.shape[0], 4)
    scores = torch.zeros(num, priors.shape[0], conf.shape[2] + 1)
    for i in range(num):
        boxes[i] = decode(loc[i], priors, variances)
        cs = obj[i, :, 1].unsqueeze(1).expand_as(conf[i]).mul(conf[i])
        scores[i] = torch.cat((obj[i, :, 0].unsqueeze(1), cs), 1)
    return boxes, scores


def box_utils_nms(boxes, scores, overlap=0.5, top_k=200):
    """utils/box_utils.py:238-302: greedy NMS over the top_k highest scores, no +1,
    keeps IoU <= overlap; returns (keep LongTensor[N] zero-padded, count).

    Tie order follows ``scores.sort(0)`` (ascending, torch's stable=False default on
    CPU is deterministic); union is evaluated as (rem_areas - inter) + area[i]."""
    n = scores.shape[0]
    keep = torch.zeros(n, dtype=torch.long)
    if boxes.numel() == 0:
        return keep, 0
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    area = (x2 - x1) * (y2 - y1)
    _, idx = scores.sort(0)
    idx = idx[-top_k:]
    count = 0
    while idx.numel() > 0:
        i = idx[-1]
        keep[count] = i
        count += 1
        if idx.numel() == 1:
            break
        idx = idx[:-1]
        xx1 = torch.clamp(x1[idx], min=float(x1[i]))
        yy1 = torch.clamp(y1[idx], min=float(y1[i]))
        xx2 = torch.clamp(x2[idx], max=float(x2[i]))
        yy2 = torch.clamp(y2[idx], max=float(y2[i]))
        w = torch.clamp(xx2 - xx1, min=0.0)
        h = torch.clamp(yy2 - yy1, min=0.0)
        inter = w * h
        union = (area[idx] - inter) + area[i]
        iou = inter / union
        idx = idx[iou.le(overlap)]
    return keep, count
