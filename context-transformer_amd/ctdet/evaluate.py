"""Detection results format + PASCAL VOC evaluation for the batched pipeline.

Restates, on in-memory arrays, what the reference does through files:
  test.py:107-109,171-172           all_boxes[class][image] = float32 [k,5]; `detections.pkl`
  data/voc0712.py:351-376           comp4_det_test_<cls>.txt lines `id score x1+1 y1+1 x2+1 y2+1`
                                     formatted {:.3f} / {:.1f}  (the rounding is part of the metric)
  data/voc_eval.py:33-66            voc_ap  (VOC07 11-point and area-under-curve)
  data/voc_eval.py:67-203           voc_eval (greedy matching at IoU > 0.5 with the +1 pixel
                                     convention, `difficult` boxes ignored, duplicates = FP)
Host-side numpy like the reference; nothing here is on the device hot path.
"""
import os
import pickle

import numpy as np


def to_reference_all_boxes(per_image):
    """[img][cls] (DetectionPipeline.results()) -> the reference's all_boxes[cls][img]."""
    ncls = len(per_image[0]) if per_image else 0
    return [[per_image[i][j] for i in range(len(per_image))] for j in range(ncls)]


def save_detections(all_boxes, path):
    """test.py:171-172: pickle of all_boxes with HIGHEST_PROTOCOL."""
    with open(path, 'wb') as f:
        pickle.dump(all_boxes, f, pickle.HIGHEST_PROTOCOL)


def results_lines(all_boxes_cls, image_ids):
    """Text lines of one class's results file (data/voc0712.py:360-376)."""
    lines = []
    for im_ind, index in enumerate(image_ids):
        dets = all_boxes_cls[im_ind]
        if len(dets) == 0:
            continue
        for k in range(dets.shape[0]):
            lines.append('{:s} {:.3f} {:.1f} {:.1f} {:.1f} {:.1f}'.format(
                index, dets[k, -1], dets[k, 0] + 1, dets[k, 1] + 1, dets[k, 2] + 1, dets[k, 3] + 1))
    return lines


def write_voc_results(all_boxes, image_ids, classes, out_dir, template='comp4_det_test_{:s}.txt'):
    os.makedirs(out_dir, exist_ok=True)
    paths = {}
    for cls_ind, cls in enumerate(classes):
        if cls == '__background__':
            continue
        paths[cls] = os.path.join(out_dir, template.format(cls))
        with open(paths[cls], 'wt') as f:
            for line in results_lines(all_boxes[cls_ind], image_ids):
                f.write(line + '\n')
    return paths


def voc_ap(rec, prec, use_07_metric=False):
    """data/voc_eval.py:33-66."""
    if use_07_metric:
        ap = 0.
        for t in np.arange(0., 1.1, 0.1):
            p = 0 if np.sum(rec >= t) == 0 else np.max(prec[rec >= t])
            ap = ap + p / 11.
        return ap
    mrec = np.concatenate(([0.], rec, [1.]))
    mpre = np.concatenate(([0.], prec, [0.]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


def voc_eval_lines(lines, gt, ovthresh=0.5, use_07_metric=False):
    """data/voc_eval.py:134-203 on the parsed lines of one class.

    gt: {image_id: {'bbox': int array [k,4], 'difficult': bool array [k]}} for THIS class
    (images without objects of the class may be missing).  Returns (rec, prec, ap)."""
    per_image, num_pos = {}, 0
    for img, r in gt.items():
        bbox = np.asarray(r['bbox']).reshape(-1, 4) if len(r['bbox']) else np.zeros((0, 4))
        difficult = np.asarray(r['difficult'], dtype=bool).reshape(-1)
        per_image[img] = {'bbox': bbox, 'difficult': difficult, 'det': [False] * len(difficult)}
        num_pos += int(np.sum(~difficult))
    split = [x.strip().split(' ') for x in lines]
    image_ids = [x[0] for x in split]
    scores = np.array([float(x[1]) for x in split])
    BB = np.array([[float(z) for z in x[2:]] for x in split])
    order = np.argsort(-scores)
    BB = BB[order, :] if BB.size != 0 else BB
    image_ids = [image_ids[x] for x in order]
    nd = len(image_ids)
    tp, fp = np.zeros(nd), np.zeros(nd)
    empty = {'bbox': np.zeros((0, 4)), 'difficult': np.zeros(0, bool), 'det': []}
    for d in range(nd):
        rec_i = per_image.get(image_ids[d], empty)
        bb = BB[d, :].astype(float)
        best_iou = -np.inf
        gt_boxes = rec_i['bbox'].astype(float)
        if gt_boxes.size > 0:
            iw = np.maximum(np.minimum(gt_boxes[:, 2], bb[2]) - np.maximum(gt_boxes[:, 0], bb[0]) + 1., 0.)
            ih = np.maximum(np.minimum(gt_boxes[:, 3], bb[3]) - np.maximum(gt_boxes[:, 1], bb[1]) + 1., 0.)
            inter_area = iw * ih
            union_area = ((bb[2] - bb[0] + 1.) * (bb[3] - bb[1] + 1.) +
                   (gt_boxes[:, 2] - gt_boxes[:, 0] + 1.) * (gt_boxes[:, 3] - gt_boxes[:, 1] + 1.) - inter_area)
            ious = inter_area / union_area
            best_iou = np.max(ious)
            best_j = np.argmax(ious)
        if best_iou > ovthresh:
            if not rec_i['difficult'][best_j]:
                if not rec_i['det'][best_j]:
                    tp[d] = 1.
                    rec_i['det'][best_j] = 1
                else:
                    fp[d] = 1.
        else:
            fp[d] = 1.
    fp, tp = np.cumsum(fp), np.cumsum(tp)
    rec = tp / float(num_pos)
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    return rec, prec, voc_ap(rec, prec, use_07_metric)


def evaluate_detections(all_boxes, image_ids, gt_by_class, classes, use_07_metric=True, ovthresh=0.5):
    """data/voc0712.py:339-426 without the files: all_boxes[cls][img] -> {cls: ap}, mean AP.
    gt_by_class[cls] is the `gt` mapping of voc_eval_lines.  Detections pass through the same text
    formatting as the results files, so scores are compared at 3 and boxes at 1 decimal."""
    aps = {}
    for cls_ind, cls in enumerate(classes):
        if cls == '__background__':
            continue
        lines = results_lines(all_boxes[cls_ind], image_ids)
        aps[cls] = float(voc_eval_lines(lines, gt_by_class.get(cls, {}), ovthresh, use_07_metric)[2])
    return aps, float(np.mean(list(aps.values()))) if aps else float('nan')


def coco_results(all_boxes, image_ids, category_ids):
    """data/coco.py:242-270: all_boxes[cls][img] -> the list of COCO result dicts
    {image_id, category_id, bbox [x, y, w, h] with the +1 width/height convention, score}.
    category_ids[cls] is the COCO id of class index cls (index 0 = background is skipped)."""
    out = []
    for cls_ind, cat_id in enumerate(category_ids):
        if cls_ind == 0:
            continue
        for im_ind, image_id in enumerate(image_ids):
            dets = all_boxes[cls_ind][im_ind]
            if len(dets) == 0:
                continue
            dets = np.asarray(dets, dtype=np.float64)
            xs, ys = dets[:, 0], dets[:, 1]
            ws, hs = dets[:, 2] - xs + 1, dets[:, 3] - ys + 1
            out.extend({'image_id': image_id, 'category_id': cat_id,
                        'bbox': [float(xs[k]), float(ys[k]), float(ws[k]), float(hs[k])],
                        'score': float(dets[k, -1])} for k in range(dets.shape[0]))
    return out


def write_coco_results(all_boxes, image_ids, category_ids, res_file):
    """data/coco.py:260-274: JSON file for pycocotools' loadRes."""
    import json
    with open(res_file, 'w') as f:
        json.dump(coco_results(all_boxes, image_ids, category_ids), f)
    return res_file


def detection_collate(batch):
    """data/voc0712.py:429-451: (image tensor, [G,6] annotation array) samples -> (stacked images, list of
    float target tensors [x1,y1,x2,y2,label,weight]) -- the layout MultiBoxLoss_combined and init_reweight take."""
    import torch
    imgs, targets = [], []
    for sample in batch:
        for item in sample:
            if torch.is_tensor(item):
                imgs.append(item)
            elif isinstance(item, np.ndarray):
                targets.append(torch.from_numpy(item).float())
    return torch.stack(imgs, 0), targets
