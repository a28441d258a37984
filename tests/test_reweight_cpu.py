"""`init_reweight` (train.py:252-286): the oracle against a golden replayed on the reference's own `match`, and
the host-side pieces of the product (class sums, 2-rank reduction)."""
import os

import numpy as np
import torch

from ctdet import reweight
from oracle import box_ref, reweight_ref

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'reweight.npz'))


def _inputs(tag):
    C = 60 if tag == 'transfer' else 15
    g = torch.Generator().manual_seed(31 if tag == 'transfer' else 32)
    P = box_ref.prior_box(box_ref.ANCHOR_CFGS['VOC_300']).shape[0]
    confs, tgts = [], []
    for it in range(2):
        confs.append(torch.randn(3, P, C, generator=g))
        for b in range(3):                       # the generator also produced the boxes: keep the stream aligned
            torch.rand(7, 2, generator=g)
            torch.rand(7, 2, generator=g)
        tgts.append([torch.from_numpy(t) for t in G['%s_targets_%d' % (tag, it)]])
    return confs, tgts


def test_oracle_matches_reference_replay():
    priors = box_ref.prior_box(box_ref.ANCHOR_CFGS['VOC_300'])
    for tag in ('transfer', 'incre'):
        confs, tgts = _inputs(tag)
        w = reweight_ref.init_reweight(confs, tgts, priors, 21, 0.5, tag)
        assert w.shape == G[tag + '_weight'].shape
        np.testing.assert_allclose(w.numpy(), G[tag + '_weight'], rtol=0, atol=2e-6)
        assert (G[tag + '_counts'] > 0).all()


def test_class_sums_equal_oracle_means():
    priors = box_ref.prior_box(box_ref.ANCHOR_CFGS['VOC_300'])
    confs, tgts = _inputs('transfer')
    sums = counts = None
    for conf, t in zip(confs, tgts):
        labels = torch.stack([box_ref.match(0.5, x[:, :-2], priors, [0.1, 0.2], x[:, -2:])[1][:, 0] for x in t])
        s, c = reweight.class_feature_sums(conf, labels, 21)
        sums, counts = (s, c) if sums is None else (sums + s, counts + c)
    assert counts.long().tolist() == G['transfer_counts'].tolist()
    w = reweight.weights_from_sums(sums, counts, 'transfer')
    np.testing.assert_allclose(w.numpy(), G['transfer_weight'], rtol=0, atol=3e-6)
    wi = reweight.weights_from_sums(sums, counts, 'incre')
    assert wi.shape[0] == 5 and torch.allclose(wi, w[15:])
