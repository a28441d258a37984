#!/usr/bin/env python3
"""The RFB blocks' ConvLinear layers (1x1, out = relu(bn(conv(x)) * scale + shortcut)) with and without their shortcut, bs 32:
what the residual read costs the bf16x3 kernel's epilogue.   python tools/res_probe.py"""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'context-transformer_amd')); sys.path.insert(0, REPO)
from ctdet import engine
DEV, B = 'cuda:0', 32
be = engine.HipBackend(DEV)
for name, Cin, H, Cout in (('Norm.linear', 512, 38, 512), ('extras.0.linear', 768, 19, 1024), ('extras.1.linear', 768, 10, 512)):
    g = torch.Generator().manual_seed(1)
    w = torch.nn.Parameter((torch.randn(Cout, Cin, 1, 1, generator=g) * (2.0 / Cin) ** 0.5).to(DEV), requires_grad=False)
    bn = torch.nn.BatchNorm2d(Cout).to(DEV).eval()
    for res in (None, 'r'):
        st = engine.ConvStep(name, [engine.ConvPart(w, None, bn, True)], Cin, 1, 1, 1, 0, 0, 1, 'x', 0, H, H, 'y', 0,
                             res=res, res_coff=448 if res else 0, res_scale=0.1)
        bufs = {'x': torch.relu(torch.randn(B, Cin, H, H, generator=g)).to(DEV), 'y': torch.empty(B, Cout, H, H, device=DEV),
                'r': torch.randn(B, 448 + Cout, H, H, generator=g).to(DEV)}
        be.prepare_conv(st, bufs, B)
        be.pack_conv(st)
        names = be.x3_names()
        for cfg in os.environ.get('CFGS', 'x3:128x128k16d,x3:64x128k16d').split(','):
            be.enable_x3(st, names.index(cfg))
            for _ in range(2):
                be.run_conv(st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                be.run_conv(st)
            e1.record(); torch.cuda.synchronize()
            print('%-16s %4d->%-4d @%2dx%-2d %-16s %s  %7.1f us' % (name, Cin, Cout, H, H, cfg, 'shortcut' if res else 'plain   ',
                                                                   e0.elapsed_time(e1) / 10 * 1e3), flush=True)
