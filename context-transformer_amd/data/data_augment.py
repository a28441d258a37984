"""Test-time input transform with the reference's name and call protocol.

data/data_augment.py:224-266: `BaseTransform(resize, rgb_means, swap)(img)` -> float32 CHW
tensor of the resized, mean-subtracted image.  Here the resize runs on the device
(`ct_preproc_resize`, one launch per batch) and the result stays there; `batch()` is the
entry the batched harness uses.  Training-time augmentation (`preproc`) stays the reference's.
"""
import torch

from ctdet import ops


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
        if target is not None:
            raise NotImplementedError('BaseTransform: the target branch (data_augment.py:247-254) is unused by '
                                      'test.py and not provided')
        return self.batch([img])[0]
