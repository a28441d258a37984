"""Deterministic synthetic weights / inputs for parity tests and bench.py.

There is no network for checkpoints or datasets, so every parity case and the bench use
a *name-seeded* filler: each state-dict entry is generated from a torch.Generator
seeded with crc32(key), which makes the same weights reproducible on the GPU box, in
the oracle and in the golden generator without shipping 147 MB.
"""
import zlib

import numpy as np
import torch


def _gen(key):
    g = torch.Generator()
    g.manual_seed(zlib.crc32(key.encode()))
    return g


def fill_tensor(key, shape):
    """One synthetic fp32 tensor for state-dict entry `key` of `shape`."""
    g = _gen(key)
    shape = tuple(shape)
    leaf = key.split('.')[-1]
    if key == 'scale':
        return torch.full(shape, 5.0)
    if key == 'Wz':
        return torch.randn(shape, generator=g) * 0.1
    if leaf == 'num_batches_tracked':
        return torch.zeros(shape, dtype=torch.long)
    if leaf == 'running_mean':
        return torch.rand(shape, generator=g) * 0.2 - 0.1
    if leaf == 'running_var':
        return torch.rand(shape, generator=g) * 0.4 + 0.8
    if '.bn.' in key and leaf == 'weight':
        return torch.rand(shape, generator=g) * 0.4 + 0.8
    if '.bn.' in key and leaf == 'bias':
        return torch.rand(shape, generator=g) * 0.2 - 0.1
    if leaf == 'bias':
        return torch.rand(shape, generator=g) * 0.1 - 0.05
    if leaf == 'weight':
        fan_out = shape[0] * int(np.prod(shape[2:])) if len(shape) > 2 else shape[0]
        w = torch.randn(shape, generator=g) * float(np.sqrt(2.0 / fan_out))
        if key == 'OBJ_Target.weight':
            w = w / w.norm(dim=1, keepdim=True)
        return w
    raise KeyError('no synthetic rule for state-dict key %r' % key)


def fill_state_dict(shapes):
    """shapes: {key: shape} (or a state_dict) -> {key: tensor}."""
    out = {}
    for k, v in shapes.items():
        shp = tuple(v.shape) if hasattr(v, 'shape') else tuple(v)
        out[k] = fill_tensor(k, shp)
    return out


def images(batch, size, kind='randn', seed=1234):
    """Synthetic input batch [B,3,S,S] fp32: 'randn' or image-like 'u8' minus BGR means."""
    g = torch.Generator()
    g.manual_seed(seed)
    if kind == 'randn':
        return torch.randn(batch, 3, size, size, generator=g)
    x = torch.randint(0, 256, (batch, 3, size, size), generator=g).float()
    return x - torch.tensor([104.0, 117.0, 123.0]).view(1, 3, 1, 1)


def clustered_dets(n, w=500.0, h=375.0, clusters=8, seed=4321, rng=None):
    """'Trained-like' detections for one (image, class): n boxes in pixel coords from a
    few clusters, tie-free scores (SURVEY 8d regime R2).  Returns float32 [n,5]."""
    rng = rng or np.random.RandomState(seed)
    cx = rng.uniform(0, 1, clusters) * w
    cy = rng.uniform(0, 1, clusters) * h
    sz = rng.uniform(20, 200, (clusters, 2))
    which = rng.randint(0, clusters, n)
    c = np.stack([cx[which], cy[which]], 1) + rng.normal(0, 6, (n, 2))
    wh = sz[which] + rng.normal(0, 6, (n, 2))
    wh = np.maximum(wh, 2.0)
    scores = rng.permutation(np.linspace(0.011, 0.99, n))
    dets = np.concatenate([c - wh / 2, c + wh / 2, scores[:, None]], 1)
    return dets.astype(np.float32)


def targets(batch, num_classes, seed=99):
    """Synthetic ground truth: list of [G,6] = [x1,y1,x2,y2,label,weight] (G ~ U{1..8})."""
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(batch):
        gcount = rng.randint(1, 9)
        xy = rng.uniform(0, 0.5, (gcount, 2))
        wh = rng.uniform(0.1, 0.5, (gcount, 2))
        lab = rng.randint(1, num_classes, (gcount, 1)).astype(np.float64)
        t = np.concatenate([xy, xy + wh, lab, np.ones((gcount, 1))], 1)
        out.append(torch.from_numpy(t.astype(np.float32)))
    return out
