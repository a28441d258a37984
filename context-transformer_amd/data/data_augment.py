"""Input transforms with the reference's names and call protocol (data/data_augment.py).

`BaseTransform(resize, rgb_means, swap)(img)` (:224-266, test time): float32 CHW tensor of the resized,
mean-subtracted image; the resize runs on the device (`ct_preproc_resize`, one launch per batch).

`preproc(resize, rgb_means, p)(image, targets[, cls])` (:164-221, training time): random crop -> photometric
distortion -> expand -> mirror -> resize -> minus means, returning (CHW tensor, normalised targets).  Split the
MI355X way: the random DECISIONS are the reference's scalar host logic -- drawn here with the same `random` calls in
the same order, so a seeded run makes the same choices -- and produce a 96-byte plan per image; the PIXELS of a whole
batch are then produced by one gather kernel (`ct_preproc_augment`) that never materialises the cropped / distorted /
expanded intermediates.  `preproc.batch(images, targets)` is the batched entry that keeps 8 GPUs fed;
`mixup_targets` / `mixup_images` are the mixup of data/voc0712.py:240-275.
"""
import math
import random

import numpy as np
import torch

from ctdet import ops
from utils.box_utils import matrix_iou

CROP_MODES = (None, (0.1, None), (0.3, None), (0.5, None), (0.7, None), (0.9, None), (None, None))
# preproc_for_test draws one of cv2's [LINEAR, CUBIC, AREA, NEAREST, LANCZOS4]; the device kernel has
# linear (0), nearest (1) and area (2) -- the two higher-order filters run as linear (augmentation noise either way)
INTERP_OF_DRAW = (0, 0, 2, 1, 0)


class BaseTransform(object):
    def __init__(self, resize, rgb_means, swap=(2, 0, 1), device='cuda', max_batch=32):
        if tuple(swap) != (2, 0, 1):
            raise ValueError('BaseTransform: only the HWC->CHW swap (2, 0, 1) is supported')
        self.means, self.resize, self.swap = rgb_means, resize, swap
        self.device, self.max_batch = torch.device(device), max_batch
        self._pre = None

    def _preprocessor(self):
        if self._pre is None:
            self._pre = ops.Preprocessor(self.resize, self.means, self.device, self.max_batch)
        return self._pre

    def batch(self, images, out=None):
        """list of uint8 HxWx3 arrays -> float32 [len,3,resize,resize] on the device."""
        return self._preprocessor()(images, out)

    def __call__(self, img, target=None):
        """data_augment.py:247-266.  With a target the reference returns (tensor, target) untouched."""
        t = self.batch([img])[0]
        return t if target is None else (t, target)


def _plan(h, w):
    """Identity plan for an h x w image: no crop, no distortion, no canvas, no mirror, linear resize."""
    return dict(H=h, W=w, crop=(0, 0, w, h), exp=(w, h, 0, 0), mirror=0, interp=0, flags=0, hue=0,
                beta=0.0, alpha=1.0, sat=1.0)


class preproc(object):
    def __init__(self, resize, rgb_means, p, device='cuda', max_batch=32, rng=None):
        self.means, self.resize, self.p = rgb_means, resize, p
        self.device, self.max_batch = torch.device(device), max_batch
        self.rng = rng or random          # the reference uses the module-level generator
        self._aug = None

    # ------------------------------------------------------------------ decisions (host scalars, no pixels)
    def _crop(self, width, height, boxes, labels, cls):
        """data_augment.py:19-79: -> (l, t, w, h), boxes, labels."""
        rng = self.rng
        if len(boxes) == 0:
            return (0, 0, width, height), boxes, labels
        while True:
            mode = rng.choice(CROP_MODES)
            if mode is None:
                return (0, 0, width, height), boxes, labels
            min_iou = float('-inf') if mode[0] is None else mode[0]
            max_iou = float('inf') if mode[1] is None else mode[1]
            for _ in range(50):
                scale = rng.uniform(0.3, 1.)
                lo, hi = max(0.5, scale * scale), min(2, 1. / scale / scale)
                ratio = math.sqrt(rng.uniform(lo, hi))
                w, h = int(scale * ratio * width), int((scale / ratio) * height)
                l, t = rng.randrange(width - w), rng.randrange(height - h)
                roi = np.array((l, t, l + w, t + h))
                iou = matrix_iou(boxes, roi[np.newaxis])
                if not (min_iou <= iou.min() and iou.max() <= max_iou):
                    continue
                centers = (boxes[:, :2] + boxes[:, 2:]) / 2
                inside = np.logical_and(roi[:2] < centers, centers < roi[2:]).all(axis=1)
                bt, lt = boxes[inside].copy(), labels[inside].copy()
                if len(bt) == 0 or (cls is not None and (lt != (cls + 1)).all()):
                    continue
                bt[:, :2] = np.maximum(bt[:, :2], roi[:2]) - roi[:2]
                bt[:, 2:] = np.minimum(bt[:, 2:], roi[2:]) - roi[:2]
                return (l, t, w, h), bt, lt

    def _distort(self, plan):
        """data_augment.py:82-110: which of brightness / contrast / hue / saturation, and by how much."""
        rng = self.rng
        if rng.randrange(2):
            plan['flags'] |= 1
            plan['beta'] = rng.uniform(-32, 32)
        if rng.randrange(2):
            plan['flags'] |= 2
            plan['alpha'] = rng.uniform(0.5, 1.5)
        if rng.randrange(2):
            plan['flags'] |= 4
            plan['hue'] = rng.randint(-18, 18)
        if rng.randrange(2):
            plan['flags'] |= 8
            plan['sat'] = rng.uniform(0.5, 1.5)

    def _expand(self, width, height, boxes):
        """data_augment.py:113-146: -> (canvas w, canvas h, left, top), boxes."""
        rng = self.rng
        if rng.random() > self.p:
            return (width, height, 0, 0), boxes
        while True:
            scale = rng.uniform(1, 4)
            lo, hi = max(0.5, 1. / scale / scale), min(2, scale * scale)
            ratio = math.sqrt(rng.uniform(lo, hi))
            ws, hs = scale * ratio, scale / ratio
            if ws < 1 or hs < 1:
                continue
            w, h = int(ws * width), int(hs * height)
            left, top = rng.randint(0, w - width), rng.randint(0, h - height)
            bt = boxes.copy()
            bt[:, :2] += (left, top)
            bt[:, 2:] += (left, top)
            return (w, h, left, top), bt

    def decide(self, shape, targets, cls=None):
        """All random choices of one `preproc.__call__` for an image of `shape` (h, w, 3) and pixel targets
        [G,5] = (x1, y1, x2, y2, label)  ->  (plan, targets_out [G',5] normalised).  Pure host logic."""
        rng = self.rng
        height_o, width_o = int(shape[0]), int(shape[1])
        targets = np.asarray(targets)
        boxes, labels = targets[:, :-1].copy(), targets[:, -1].copy()
        targets_o = targets.copy()                       # fallback: the untouched image with normalised boxes
        targets_o[:, 0:4:2] /= width_o
        targets_o[:, 1:4:2] /= height_o
        plan = _plan(height_o, width_o)
        plan['crop'], boxes, labels = self._crop(width_o, height_o, boxes, labels, cls)
        self._distort(plan)
        cw, ch = plan['crop'][2], plan['crop'][3]
        plan['exp'], boxes = self._expand(cw, ch, boxes)
        width, height = plan['exp'][0], plan['exp'][1]
        if rng.randrange(2):                             # _mirror (:149-155)
            plan['mirror'] = 1
            boxes = boxes.copy()
            boxes[:, 0::2] = width - boxes[:, 2::-2]
        plan['interp'] = INTERP_OF_DRAW[rng.randrange(5)]            # preproc_for_test (:158-161)
        boxes = boxes.copy()
        boxes[:, 0::2] /= width
        boxes[:, 1::2] /= height
        keep = np.minimum(boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]) > 0.01
        bt, lt = boxes[keep], labels[keep].copy()
        if len(bt) == 0 or (cls is not None and (lt != (cls + 1)).all()):
            plan = _plan(height_o, width_o)              # the original image, resized (:205-212)
            plan['interp'] = INTERP_OF_DRAW[rng.randrange(5)]
            return plan, targets_o
        return plan, np.hstack((bt, np.expand_dims(lt, 1)))

    # ------------------------------------------------------------------ pixels (device)
    def _augmenter(self):
        if self._aug is None:
            self._aug = ops.Augmenter(self.resize, self.means, self.device, self.max_batch)
        return self._aug

    def batch(self, images, targets, cls=None, out=None):
        """lists of uint8 HxWx3 images and [G,5] pixel targets -> (float32 [n,3,S,S] on the device, [targets_out])."""
        plans, touts = [], []
        for img, tg in zip(images, targets):
            plan, tout = self.decide(img.shape, tg, cls)
            plans.append(plan)
            touts.append(tout)
        return self._augmenter()(images, plans, out), touts

    def __call__(self, image, targets, cls=None):
        imgs, touts = self.batch([image], [targets], cls)
        return imgs[0], touts[0]


def mixup_targets(target1, target2, lambd, ignore_minus1=False):
    """data/voc0712.py:263-273: [G1,5] + [G2,5] -> [G1+G2,6] with the mixup weight as the last column; in phase 2 of
    the incremental setting boxes labelled -1 get weight 0.  lambd >= 1: target1 with weight 1 (:247-250)."""
    if target2 is None or lambd >= 1:
        return np.hstack((target1, np.ones((target1.shape[0], 1))))
    y1 = np.hstack((target1, np.full((target1.shape[0], 1), float(lambd))))
    y2 = np.hstack((target2, np.full((target2.shape[0], 1), 1. - float(lambd))))
    mix = np.vstack((y1, y2))
    if ignore_minus1:
        mix[mix[:, -2] == -1, -1] = 0
    return mix


def mixup_images(img1, img2, lambd):
    """data/voc0712.py:262 on device batches: img1 * lambd + img2 * (1 - lambd), lambd a float or one per image."""
    return ops.mixup_blend(img1, img2, lambd)
