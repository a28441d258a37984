#!/usr/bin/env python3
"""One conv layer through one kernel variant, for rocprofv3 counter passes:
    python tools/x3_one.py --layer conv7 --x3 11 [--iters 10]      (--x3 -1: the fp32 MFMA kernel, best table tile)"""
import argparse, os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'context-transformer_amd')); sys.path.insert(0, REPO)
from ctdet import _lib, engine
LAYERS = {  # Cin, H, W, Cout, k, stride, pad, dil
    'conv6': (512, 19, 19, 1024, 3, 1, 6, 6), 'conv7': (1024, 19, 19, 1024, 1, 1, 0, 1),
    'norm_reduce': (512, 38, 38, 960, 1, 1, 0, 1), 'norm_d3': (128, 38, 38, 128, 3, 1, 3, 3),
}
ap = argparse.ArgumentParser()
ap.add_argument('--layer', default='conv7'); ap.add_argument('--x3', type=int, default=11)
ap.add_argument('--f32cfg', type=int, default=8); ap.add_argument('--iters', type=int, default=10); ap.add_argument('--batch', type=int, default=32)
a = ap.parse_args()
Cin, H, W, Cout, k, stride, pad, dil = LAYERS[a.layer]
be = engine.HipBackend('cuda:0')
g = torch.Generator().manual_seed(1)
w = torch.nn.Parameter((torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5).cuda(), requires_grad=False)
st = engine.ConvStep(a.layer, [engine.ConvPart(w, None, None, False)], Cin, k, k, stride, pad, pad, dil, 'x', 0, H, W, 'y', 0)
bufs = {'x': torch.relu(torch.randn(a.batch, Cin, H, W, generator=g)).cuda(), 'y': torch.empty(a.batch, Cout, st.oh, st.ow, device='cuda')}
be.prepare_conv(st, bufs, a.batch)
if a.x3 >= 0:
    be.enable_x3(st, a.x3)
else:
    st.rt['desc'].config = a.f32cfg
for _ in range(a.iters):
    be.run_conv(st)
torch.cuda.synchronize()
print('done', a.layer, a.x3)
