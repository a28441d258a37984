#!/usr/bin/env python3
"""Error map of one Winograd variant on a small case (development aid): python tools/wino_debug.py TILE B Cin H W Cout"""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'context-transformer_amd')); sys.path.insert(0, REPO)
from ctdet import engine
tile, B, Cin, H, W, Cout = [int(v) for v in sys.argv[1:7]]
torch.manual_seed(0)
DEV = 'cuda:0'
be = engine.HipBackend(DEV)
w = torch.nn.Parameter(torch.randn(Cout, Cin, 3, 3, device=DEV) * 0.1, requires_grad=False)
b = torch.nn.Parameter(torch.zeros(Cout, device=DEV), requires_grad=False)
st = engine.ConvStep('t', [engine.ConvPart(w, b, None, False)], Cin, 3, 3, 1, 1, 1, 1, 'x', 0, H, W, 'y', 0)
bufs = {'x': torch.randn(B, Cin, H, W, device=DEV), 'y': torch.full((B, Cout, H, W), float('nan'), device=DEV)}
be.prepare_conv(st, bufs, B)
be.enable_wino(st, tile=tile)
be.run_conv(st)
torch.cuda.synchronize()
ref = torch.nn.functional.conv2d(bufs['x'].double(), w.double(), None, padding=1)
err = (bufs['y'].double() - ref).abs() / ref.abs().max()
print('max err %.3e, nan %d' % (err.nan_to_num(9).max().item(), torch.isnan(bufs['y']).sum().item()))
torch.set_printoptions(linewidth=250, precision=1, sci_mode=True)
print('per pixel (max over n, cout):'); print(err.nan_to_num(9).amax((0, 1)))
print('per cout (max):'); print(err.nan_to_num(9).amax((0, 2, 3)))
print('per image:'); print(err.nan_to_num(9).amax((1, 2, 3)))
# which input channels matter: zero all but one channel group
for g in range(0, Cin, 4):
    x2 = torch.zeros_like(bufs['x']); x2[:, g:g + 4] = bufs['x'][:, g:g + 4]
    bufs2 = {'x': x2, 'y': torch.full((B, Cout, H, W), float('nan'), device=DEV)}
    st2 = engine.ConvStep('t', [engine.ConvPart(w, b, None, False)], Cin, 3, 3, 1, 1, 1, 1, 'x', 0, H, W, 'y', 0)
    be.prepare_conv(st2, bufs2, B); be.enable_wino(st2, tile=tile); be.run_conv(st2); torch.cuda.synchronize()
    r2 = torch.nn.functional.conv2d(x2.double(), w.double(), None, padding=1)
    print('channels %2d..%2d only: max err %.2e' % (g, g + 3, ((bufs2['y'].double() - r2).abs().max() / r2.abs().max()).item()))
