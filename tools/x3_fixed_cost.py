import os, sys, torch
sys.path.insert(0, '/root/repo/context-transformer_amd'); sys.path.insert(0, '/root/repo')
from ctdet import _lib, engine
be = engine.HipBackend('cuda:0'); lib = _lib.lib()
def t(Cin, Cout, H, B=32, k=1, cfg=0, n=10):
    g = torch.Generator().manual_seed(1)
    w = torch.nn.Parameter(torch.randn(Cout, Cin, k, k, generator=g).cuda() * 0.05, requires_grad=False)
    st = engine.ConvStep('t', [engine.ConvPart(w, None, None, False)], Cin, k, k, 1, k // 2, k // 2, 1, 'x', 0, H, H, 'y', 0)
    bufs = {'x': torch.randn(B, Cin, H, H, generator=g).cuda(), 'y': torch.empty(B, Cout, H, H, device='cuda')}
    be.prepare_conv(st, bufs, B)
    st.rt['desc'].ksplit = 0
    be.enable_x3(st, cfg)
    for _ in range(3): be.run_conv(st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): be.run_conv(st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for cfg in (0, 1):
    for Cin in (16, 32, 64, 128, 256, 512, 1024):
        us = t(Cin, 1024, 19, cfg=cfg)
        print('cfg %d  1x1 %4d->1024 @19x19 bs32: %7.1f us  (%d k-steps of 16)' % (cfg, Cin, us, Cin // 16))
