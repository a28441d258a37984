#!/usr/bin/env python3
"""Time the Context-Transformer block alone (ct_ctx_attention_fwd) at the BASELINE shapes."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'context-transformer_amd')); sys.path.insert(0, REPO)
from ctdet import ops
for (B, P, M, d, T) in ((32, 11620, 1858, 60, 20), (32, 32756, 4964, 60, 20)):
    g = torch.Generator().manual_seed(0)
    conf = (torch.randn(B, P, d, generator=g) * 1.5).cuda()
    pool = (torch.randn(B, M, d, generator=g) * 1.5).cuda()
    r = lambda *s: torch.randn(*s, generator=g).cuda()
    prm = dict(theta_w=r(d, d) * 0.08, theta_b=r(d) * 0.1, phi_w=r(d, d) * 0.08, phi_b=r(d) * 0.1, g_w=r(d, d) * 0.08,
               g_b=r(d) * 0.1, wz=r(d) * 0.5, obj_w=r(T, d) * 0.3, scale=5.0)
    for _ in range(3):
        ops.ctx_attention(conf, pool, prm, False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.ctx_attention(conf, pool, prm, False)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 4.0 * P * M * 64 * B            # padded feature dim is what the MFMAs execute
    print('B=%d P=%d M=%d: %.3f ms  (%.1f TFLOP/s executed, %.1f algorithmic d=%d)' % (B, P, M, ms, fl / ms / 1e9, fl * d / 64 / ms / 1e9, d))
