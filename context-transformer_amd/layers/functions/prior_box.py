"""Prior (anchor) boxes -- drop-in for the reference's layers/functions/prior_box.py.

Computed once on the host in Python doubles and cast to fp32 exactly like the reference
(layers/functions/prior_box.py:31-56), so the tensor is bit-identical (sha-pinned in
tests/golden/box_ops.npz); the caller moves it to the device (test.py:89-93).
"""
from math import sqrt

import torch


class PriorBox(object):
    def __init__(self, cfg):
        self.image_size = cfg['min_dim']
        self.num_priors = len(cfg['aspect_ratios'])
        self.variance = cfg['variance'] or [0.1]
        self.feature_maps = cfg['feature_maps']
        self.min_sizes = cfg['min_sizes']
        self.max_sizes = cfg['max_sizes']
        self.steps = cfg['steps']
        self.aspect_ratios = cfg['aspect_ratios']
        self.clip = cfg['clip']
        if any(v <= 0 for v in self.variance):
            raise ValueError('Variances must be greater than 0')

    def forward(self):
        rows = []
        for level, fmap in enumerate(self.feature_maps):
            cells = self.image_size / self.steps[level]
            small = self.min_sizes[level] / self.image_size
            big = sqrt(small * (self.max_sizes[level] / self.image_size))
            shapes = [(small, small), (big, big)]
            for ar in self.aspect_ratios[level]:
                r = sqrt(ar)
                shapes += [(small * r, small / r), (small / r, small * r)]
            for i in range(fmap):
                cy = (i + 0.5) / cells
                for j in range(fmap):
                    cx = (j + 0.5) / cells
                    rows.extend((cx, cy, bw, bh) for bw, bh in shapes)
        out = torch.tensor(rows, dtype=torch.float64).to(torch.float32).view(-1, 4)
        if self.clip:
            out.clamp_(max=1, min=0)
        return out
