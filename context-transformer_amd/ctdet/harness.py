"""Batched evaluation harness: test.py:do_test (test.py:96-175) with the per-image loop replaced
by whole batches through DetectionPipeline.

Keeps the reference's dataset protocol (`len(dataset)`, `dataset.pull_image(i)` -> HxWx3 uint8,
`dataset.evaluate_detections(all_boxes, save_folder)`) and its outputs (all_boxes[cls][img],
`detections.pkl`).  The ragged last batch is padded with zero images whose results are dropped.
"""
import os

import torch

from . import evaluate
from .pipeline import DetectionPipeline


def detect_dataset(net, priors, dataset, transform, num_fg, batch=32, max_per_image=200, thresh=0.01,
                   force_cpu_rule=False, progress=None):
    """-> all_boxes[cls][img] (cls 0 = background, empty lists as in test.py:107-108)."""
    n = len(dataset)
    all_boxes = [[[] for _ in range(n)] for _ in range(num_fg + 1)]
    if n == 0:
        return all_boxes
    pipe = DetectionPipeline(net, priors, batch, num_fg, conf_thresh=thresh, max_per_image=max_per_image,
                             force_cpu_rule=force_cpu_rule)
    dev = pipe.device
    x = torch.zeros(batch, 3, net.size, net.size, device=dev)
    for start in range(0, n, batch):
        m = min(batch, n - start)
        wh = torch.ones(batch, 2)
        imgs = [dataset.pull_image(start + k) for k in range(m)]
        for k, img in enumerate(imgs):
            wh[k, 0], wh[k, 1] = img.shape[1], img.shape[0]
        if hasattr(transform, 'batch'):                 # device transform: one launch per batch
            transform.batch(imgs, out=x[:m])
        else:
            for k, img in enumerate(imgs):
                x[k].copy_(transform(img), non_blocking=True)
        if m < batch:
            x[m:].zero_()
        pipe.run(x, image_wh=wh)
        per_image = pipe.results()
        for k in range(m):
            for j in range(1, num_fg + 1):
                all_boxes[j][start + k] = per_image[k][j]
        if progress is not None:
            progress(start + m, n)
    return all_boxes


def do_test(net, priors, dataset, transform, num_fg, save_folder, batch=32, max_per_image=200, thresh=0.01,
            force_cpu_rule=False, retest=False):
    """test.py:96-175: detect, write `detections.pkl`, hand over to the dataset's evaluator."""
    import pickle
    os.makedirs(save_folder, exist_ok=True)
    det_file = os.path.join(save_folder, 'detections.pkl')
    if retest:
        with open(det_file, 'rb') as f:
            all_boxes = pickle.load(f)
    else:
        all_boxes = detect_dataset(net, priors, dataset, transform, num_fg, batch, max_per_image, thresh,
                                   force_cpu_rule)
        evaluate.save_detections(all_boxes, det_file)
    if hasattr(dataset, 'evaluate_detections'):
        return all_boxes, dataset.evaluate_detections(all_boxes, save_folder)
    return all_boxes, None
