"""Box utilities -- drop-in for the reference's utils/box_utils.py on HIP tensors.

decode / encode / jaccard / match run as hand-written HIP kernels (libctdet: ct_decode,
ct_encode, ct_jaccard, ct_match_batched) compiled without FMA contraction so they follow the
reference's fp32 rounding sequence; `match` keeps the reference's in-place signature
(utils/box_utils.py:83-132) but is one fused launch pair instead of a Python loop over ground
truths.  point_form / center_size / intersect are trivial slice arithmetic kept as tensor
expressions (not on the hot path: `match` and `jaccard` fuse point_form).  `center_size` is
syntactically broken in the reference (:25-26, raises TypeError); here it does what its
docstring says.  Device tensors only -- there is no CPU implementation.
"""
import numpy as np
import torch

from ctdet import ops


def point_form(boxes):
    """(cx,cy,w,h) -> (xmin,ymin,xmax,ymax)   [utils/box_utils.py:5-14]"""
    half = boxes[:, 2:] / 2
    return torch.cat((boxes[:, :2] - half, boxes[:, :2] + half), 1)


def center_size(boxes):
    """(xmin,ymin,xmax,ymax) -> (cx,cy,w,h)   [intent of utils/box_utils.py:17-26]"""
    return torch.cat(((boxes[:, 2:] + boxes[:, :2]) / 2, boxes[:, 2:] - boxes[:, :2]), 1)


def intersect(box_a, box_b):
    """Pairwise intersection area [A,B]   [utils/box_utils.py:29-47]"""
    hi = torch.min(box_a[:, None, 2:], box_b[None, :, 2:])
    lo = torch.max(box_a[:, None, :2], box_b[None, :, :2])
    wh = torch.clamp(hi - lo, min=0)
    return wh[:, :, 0] * wh[:, :, 1]


def jaccard(box_a, box_b):
    """Pairwise IoU of point-form boxes, [A,4] x [B,4] -> [A,B]   [utils/box_utils.py:50-68]"""
    return ops.jaccard(box_a.contiguous().float(), box_b.contiguous().float())


def matrix_iou(a, b):
    """numpy IoU used by the data augmentation   [utils/box_utils.py:70-80]"""
    lt = np.maximum(a[:, np.newaxis, :2], b[:, :2])
    rb = np.minimum(a[:, np.newaxis, 2:], b[:, 2:])
    area_i = np.prod(rb - lt, axis=2) * (lt < rb).all(axis=2)
    area_a = np.prod(a[:, 2:] - a[:, :2], axis=1)
    area_b = np.prod(b[:, 2:] - b[:, :2], axis=1)
    return area_i / (area_a[:, np.newaxis] + area_b - area_i)


def match(threshold, truths, priors, variances, labels, loc_t, conf_t, obj_t, idx, overlap=None):
    """Fill loc_t[idx] / conf_t[idx] / obj_t[idx] for one image   [utils/box_utils.py:83-132]

    truths [G,4] point form, labels [G,2] = (label, mixup weight), priors [P,4] centre form.
    Force-match collisions resolve as in the reference (later ground truth wins, :122-123)."""
    tgt = torch.cat((truths.float(), labels.float()), 1)
    res = ops.match_batched([tgt], priors.contiguous(), threshold, variances, want_overlap=overlap is not None)
    loc_t[idx] = res[0][0]
    conf_t[idx] = res[1][0]
    obj_t[idx] = res[2][0]
    if overlap is not None:
        overlap[idx] = res[3][0]


def encode(matched, priors, variances):
    """[utils/box_utils.py:135-156]"""
    return ops.encode(matched.contiguous(), priors.contiguous(), variances)


def decode(loc, priors, variances):
    """[utils/box_utils.py:184-202]"""
    return ops.decode(loc.contiguous(), priors.contiguous(), variances)


def nms(boxes, scores, overlap=0.5, top_k=200):
    """Greedy NMS over the top_k scores, no +1 convention, keeps IoU <= overlap; returns
    (keep LongTensor[N] zero padded, count)   [utils/box_utils.py:238-302]

    Runs on the device through libctdet's plain-IoU mode (union evaluated as
    (area_j - inter) + area_i like the reference)."""
    keep = torch.zeros(scores.size(0), dtype=torch.long, device=scores.device)
    if boxes.numel() == 0:
        return keep
    s = scores.detach().float().cpu().numpy()
    order = np.argsort(s, kind='stable')[-top_k:][::-1]        # descending, later index first on ties
    b = boxes.detach().float().cpu().numpy()[order]
    dets = np.concatenate([b, s[order, None]], 1).astype(np.float32)
    kept = ops.nms_sorted_host(dets, overlap, ge=False, plain_iou=True,
                               device_id=boxes.device.index or 0 if boxes.is_cuda else 0)
    idx = torch.from_numpy(order[kept].astype(np.int64))
    keep[:len(kept)] = idx.to(keep.device)
    return keep, len(kept)
