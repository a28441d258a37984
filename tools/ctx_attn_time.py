#!/usr/bin/env python3
"""Times ct_ctx_attention_fwd alone at the two production shapes (RFBNet-300 / -512 with the Context-Transformer
block, batch 32): python tools/ctx_attn_time.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'context-transformer_amd'))
from ctdet import ops  # noqa: E402

dev, d, T = 'cuda', 60, 20
g = torch.Generator().manual_seed(0)
prm = dict(wz=torch.ones(d), obj_w=torch.randn(T, d, generator=g), scale=5.0)
for k in ('theta', 'phi', 'g'):
    prm[k + '_w'] = torch.randn(d, d, generator=g) * 0.05
    prm[k + '_b'] = torch.randn(d, generator=g) * 0.1
prm = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in prm.items()}
for (B, P, M) in ((32, 11620, 1858), (32, 32756, 4964)):     # engine.Plan.M of the two networks
    conf = torch.randn(B, P, d, device=dev) * 2.7
    pool = torch.randn(B, M, d, device=dev) * 7.5
    out, ws = ops.ctx_attention_buffers(B, P, M, d, T, False, dev)
    for _ in range(3):
        ops.ctx_attention(conf, pool, prm, False, out, ws)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ops.ctx_attention(conf, pool, prm, False, out, ws)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 100
    flops = 4.0 * B * P * M * 64                    # two contractions, padded d = 64
    print('B%d P%d M%d: %.3f ms  %.1f TFLOP/s fp32-equivalent' % (B, P, M, ms, flops / ms / 1e9))
