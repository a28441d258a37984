"""Detect -- drop-in for the reference's layers/functions/detection.py.

`Detect(num_classes, bkg_label, cfg).forward((loc, conf, obj), priors)` returns
(boxes [B,P,4], scores [B,P,num_classes]) with scores = [obj0, obj1 * conf]
(layers/functions/detection.py:44-53) from ONE fused HIP kernel (decode + score fusion for
the whole batch) instead of the reference's per-image Python loop.
"""
from ctdet import ops


class Detect(object):
    def __init__(self, num_classes, bkg_label, cfg):
        self.num_classes = num_classes
        self.background_label = bkg_label
        self.variance = cfg['variance']

    def forward(self, predictions, prior):
        loc, conf, obj = predictions
        if conf.shape[-1] + 1 != self.num_classes:
            raise ValueError('Detect(num_classes=%d) got %d foreground scores' % (self.num_classes, conf.shape[-1]))
        prior = prior.to(loc.device)
        self.num_priors = prior.size(0)
        self.boxes, self.scores = ops.detect_fused(loc.contiguous(), conf.contiguous(), obj.contiguous(),
                                                   prior.contiguous(), self.variance, apply_softmax=False)
        return self.boxes, self.scores

    __call__ = forward
