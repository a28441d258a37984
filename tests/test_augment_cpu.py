"""SURVEY 8f row 4, training side: the random DECISIONS and box arithmetic of `preproc.__call__`
(data/data_augment.py:164-221) and the mixup target construction (data/voc0712.py:240-275) on the host,
against tests/golden/augment.npz -- the reference's own `preproc` run with a pixel-free cv2 stand-in
(tools/gen_goldens.py `augment`).  The pixels are device work: tests/test_gpu_augment.py."""
import hashlib
import random

import numpy as np
import pytest

from data.data_augment import preproc, mixup_targets


def test_preproc_decisions_and_targets_match_reference(golden):
    g = golden('augment.npz')
    n = int(g['ncase'])
    kinds = set()
    for k in range(n):
        h, w, cls = [int(v) for v in g['case%d_shape' % k]]
        random.seed(1000 + k)
        pre = preproc(300, (104, 117, 123), 0.6, device='cpu')
        plan, tout = pre.decide((h, w, 3), g['case%d_in' % k].copy(), None if cls < 0 else cls)
        want = g['case%d_out' % k]
        assert tout.shape == want.shape, k
        assert np.array_equal(np.asarray(tout, dtype=np.float64), want), k       # same arithmetic on the same draws
        state = np.frombuffer(hashlib.sha256(repr(random.getstate()).encode()).digest()[:8], dtype=np.uint8)
        assert np.array_equal(state, g['case%d_state' % k]), k                    # same number and order of random draws
        l, t, cw, ch = plan['crop']
        ew, eh, left, top = plan['exp']
        assert 0 <= l and 0 <= t and l + cw <= w and t + ch <= h and left + cw <= ew and top + ch <= eh
        kinds.add((plan['crop'] != (0, 0, w, h), (ew, eh) != (cw, ch), plan['mirror'], plan['flags'] != 0))
    assert len(kinds) >= 8          # the cases exercise crops, canvases, mirrors and distortions in combination


def test_mixup_targets():
    t1 = np.array([[.1, .1, .5, .5, 3.], [.2, .2, .9, .9, -1.]])
    t2 = np.array([[.3, .3, .6, .8, 7.]])
    m = mixup_targets(t1, t2, 0.3)
    assert m.shape == (3, 6) and np.allclose(m[:, -1], [0.3, 0.3, 0.7]) and np.array_equal(m[:, :5], np.vstack((t1, t2)))
    m = mixup_targets(t1, t2, 0.3, ignore_minus1=True)
    assert np.allclose(m[:, -1], [0.3, 0.0, 0.7])
    assert np.array_equal(mixup_targets(t1, t2, 1.0), np.hstack((t1, np.ones((2, 1)))))
    assert np.array_equal(mixup_targets(t1, None, 0.2), np.hstack((t1, np.ones((2, 1)))))
