"""Device pixels of the training-time augmentation (ct_preproc_augment) against oracle/augment_ref.py on seeded
images and plans drawn by the product's own decision logic, and the mixup blend.  Tolerance-based:
the oracle restates cv2's 8-bit formulas in float64, the kernel evaluates them in fp32 -- a value that lands on a
rounding boundary may differ by one grey level."""
import random

import numpy as np
import pytest
import torch

from data.data_augment import preproc, mixup_images, BaseTransform
from oracle import augment_ref

pytestmark = pytest.mark.gpu
MEANS = (104, 117, 123)


def _image(rng, h, w):
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([(xx * 3 + yy) % 256, (xx + yy * 2) % 256, (xx * yy // 7) % 256], -1)
    return ((base + rng.randint(0, 64, (h, w, 3))) % 256).astype(np.uint8)


def _targets(rng, h, w):
    G = rng.randint(1, 5)
    xy = rng.uniform(0, 0.6, (G, 2)) * (w, h)
    wh = rng.uniform(0.1, 0.4, (G, 2)) * (w, h)
    return np.hstack([xy, np.minimum(xy + wh, (w - 1, h - 1)), rng.randint(0, 20, (G, 1)).astype(np.float64)])


@pytest.mark.parametrize('size', [300, 512])
def test_augment_pixels_vs_oracle(size):
    rng = np.random.RandomState(size)
    random.seed(size)
    pre = preproc(size, MEANS, 0.6)
    imgs = [_image(rng, rng.randint(90, 260), rng.randint(90, 330)) for _ in range(24)]
    tgs = [_targets(rng, im.shape[0], im.shape[1]) for im in imgs]
    plans = []
    orig_decide = pre.decide

    def spy(shape, tg, cls=None):
        plan, out = orig_decide(shape, tg, cls)
        plans.append(plan)
        return plan, out
    pre.decide = spy
    out, touts = pre.batch(imgs, tgs)
    assert out.shape == (24, 3, size, size) and out.is_cuda and len(touts) == 24
    got = out.cpu().numpy()
    seen = set()
    for i, (im, plan) in enumerate(zip(imgs, plans)):
        want = augment_ref.augment(im, plan, size, MEANS)
        d = np.abs(got[i] - want)
        # whole grey levels; distortion + interpolation boundaries may move a few pixels by one (two through the HSV round trip)
        assert d.max() <= 2.0 and (d > 0).mean() < 0.05, (i, plan['interp'], plan['flags'], d.max(), (d > 0).mean())
        if not plan['flags'] and plan['interp'] == 1:
            assert d.max() == 0.0, (i, plan)                 # pure gather: exact
        seen.add((plan['interp'], bool(plan['flags']), plan['mirror'], plan['exp'][:2] != plan['crop'][2:]))
    assert len(seen) >= 6
    # single-image call protocol of the reference: (tensor CHW, targets)
    random.seed(3)
    t, tg = pre(imgs[0], tgs[0])
    assert t.shape == (3, size, size) and tg.shape[1] == 5


def test_identity_plan_equals_plain_resize():
    """No crop / distortion / canvas / mirror, linear: the augmentation kernel is a float bilinear resize -- within one
    grey level of the fixed-point BaseTransform path on the same image."""
    rng = np.random.RandomState(1)
    im = _image(rng, 375, 500)
    from data.data_augment import _plan
    from ctdet import ops
    aug = ops.Augmenter(300, MEANS, 'cuda')
    a = aug([im], [_plan(375, 500)])[0].cpu()
    b = BaseTransform(300, MEANS)(im).cpu()
    assert float((a - b).abs().max()) <= 1.0


def test_mixup_blend():
    g = torch.Generator().manual_seed(0)
    a, b = torch.randn(4, 3, 64, 64, generator=g).cuda(), torch.randn(4, 3, 64, 64, generator=g).cuda()
    lam = torch.tensor([0.0, 0.25, 0.5, 1.0])
    got = mixup_images(a, b, lam.cuda()).cpu()
    want = a.cpu() * lam.view(4, 1, 1, 1) + b.cpu() * (1 - lam.view(4, 1, 1, 1))
    assert torch.allclose(got, want, atol=1e-6)
    assert torch.allclose(mixup_images(a, b, 0.3).cpu(), a.cpu() * 0.3 + b.cpu() * 0.7, atol=1e-6)
