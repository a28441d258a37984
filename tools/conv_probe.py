#!/usr/bin/env python3
"""Time single conv shapes for chosen tile configs (debug / tuning aid).
   python tools/conv_probe.py            # a few VGG shapes, all configs"""
import os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'context-transformer_amd')); sys.path.insert(0, REPO)
from ctdet import _lib, engine

DEV = 'cuda:0'
SHAPES = [  # name, B, Cin, H, W, Cout, k, pad, dil
    ('base.19', 32, 512, 38, 38, 512, 3, 1, 1),
    ('base.12', 32, 256, 75, 75, 256, 3, 1, 1),
    ('base.2', 32, 64, 300, 300, 64, 3, 1, 1),
    ('base.24', 32, 512, 19, 19, 512, 3, 1, 1),
]
cfgs = [int(c) for c in os.environ.get('PROBE_CFGS', '1,2,3,4,7,8').split(',')]
be = engine.HipBackend(DEV)
lib = _lib.lib()
for (name, B, Cin, H, W, Cout, k, pad, dil) in SHAPES:
    w = torch.nn.Parameter(torch.randn(Cout, Cin, k, k, device=DEV) * 0.05, requires_grad=False)
    b = torch.nn.Parameter(torch.zeros(Cout, device=DEV), requires_grad=False)
    st = engine.ConvStep(name, [engine.ConvPart(w, b, None, True)], Cin, k, k, 1, pad, pad, dil, 'x', 0, H, W, 'y', 0)
    bufs = {'x': torch.randn(B, Cin, H, W, device=DEV), 'y': torch.empty(B, Cout, st.oh, st.ow, device=DEV)}
    be.prepare_conv(st, bufs, B)
    row = []
    for cfg in cfgs:
        st.rt['desc'].config = cfg
        for _ in range(2):
            be.run_conv(st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            be.run_conv(st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        row.append('%s %.3fms %.1fTF' % (lib.ct_conv_config_name(cfg - 1).decode(), ms, st.flops(B) / ms / 1e9))
    print('%-8s abl=%s | %s' % (name, os.environ.get('CTDET_CONV_ABLATE', '0'), ' | '.join(row)), flush=True)
