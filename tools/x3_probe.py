#!/usr/bin/env python3
"""bf16x3 (ct_conv2d_x3_fwd) vs the fp32 MFMA kernel (ct_conv2d_fwd) on the non-Winograd layers of RFBNet-300, bs 32:
time of every tile config of both, and the error of both against an fp64 convolution (first 2 images).
    python tools/x3_probe.py [--batch 32]"""
import argparse, os, sys
import torch
import torch.nn.functional as F
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'context-transformer_amd')); sys.path.insert(0, REPO)
from ctdet import _lib, engine

ap = argparse.ArgumentParser(); ap.add_argument('--batch', type=int, default=32); ap.add_argument('--noerr', action='store_true')
a = ap.parse_args()
DEV = 'cuda:0'
SHAPES = [  # name, Cin, H, W, Cout, k, stride, pad, dil
    ('conv6 3x3 d6', 512, 19, 19, 1024, 3, 1, 6, 6), ('conv7 1x1', 1024, 19, 19, 1024, 1, 1, 0, 1),
    ('Norm.reduce 1x1', 512, 38, 38, 960, 1, 1, 0, 1), ('Norm.linear 1x1', 512, 38, 38, 512, 1, 1, 0, 1),
    ('Norm 3x3 d3', 128, 38, 38, 128, 3, 1, 3, 3), ('Norm 3x1', 128, 38, 38, 128, (3, 1), 1, (1, 0), 1),
    ('ex0.reduce 1x1', 1024, 19, 19, 1536, 1, 1, 0, 1), ('ex0 3x3 d3', 256, 19, 19, 256, 3, 1, 3, 3),
    ('ex0.linear 1x1', 768, 19, 19, 1024, 1, 1, 0, 1), ('ex1.reduce2 1x1 s2', 1024, 19, 19, 768, 1, 2, 0, 1),
    ('ex1 3x3 s2', 128, 19, 19, 256, 3, 2, 1, 1), ('ex2 3x3 d2', 64, 5, 5, 64, 3, 1, 2, 2),
]
be = engine.HipBackend(DEV)
lib = _lib.lib()
B = a.batch


def timeit(st, n=5):
    for _ in range(2):
        be.run_conv(st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        be.run_conv(st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (name, Cin, H, W, Cout, k, stride, pad, dil) in SHAPES:
    kh, kw = (k, k) if isinstance(k, int) else k
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    g = torch.Generator().manual_seed(1)
    w = torch.nn.Parameter((torch.randn(Cout, Cin, kh, kw, generator=g) * (2.0 / (Cin * kh * kw)) ** 0.5).to(DEV), requires_grad=False)
    st = engine.ConvStep(name, [engine.ConvPart(w, None, None, False)], Cin, kh, kw, stride, ph, pw, dil, 'x', 0, H, W, 'y', 0)
    x = torch.relu(torch.randn(B, Cin, H, W, generator=g))
    bufs = {'x': x.to(DEV), 'y': torch.empty(B, Cout, st.oh, st.ow, device=DEV)}
    be.prepare_conv(st, bufs, B)
    want = None if a.noerr else F.conv2d(x[:2].double(), w.detach().cpu().double(), None, stride, (ph, pw), dil)

    def err():
        if want is None:
            return (0.0, 0.0)
        got = bufs['y'][:2].cpu().double()
        return (float((got - want).abs().max() / want.abs().max()), float(((got - want) ** 2).mean().sqrt() / (want ** 2).mean().sqrt()))
    f32 = []
    for cfg in range(lib.ct_conv_num_configs()):
        st.rt['desc'].config = cfg + 1
        try:
            f32.append((timeit(st), lib.ct_conv_config_name(cfg).decode()))
        except _lib.CtdetError:
            pass
    tb, nb = min(f32)
    st.rt['desc'].config = [lib.ct_conv_config_name(i).decode() for i in range(lib.ct_conv_num_configs())].index(nb) + 1
    be.run_conv(st); torch.cuda.synchronize()
    eb = err()
    x3 = []
    for cfg in range(lib.ct_conv_x3_num_configs()):
        if Cin % lib.ct_conv_x3_config_bk(cfg):
            continue
        be.enable_x3(st, cfg)
        t = timeit(st)
        x3.append((t, lib.ct_conv_x3_config_name(cfg).decode(), err()))
    fl = st.flops(B)
    print('%-20s %6.1f GFLOP | fp32 %-9s %7.1f us %6.1f TF err max %.2e rms %.2e' % (name, fl / 1e9, nb, tb * 1e3, fl / tb / 1e9, eb[0], eb[1]))
    for kind, sel in (('single acc', [r for r in x3 if not r[1].endswith('d') and 'abl' not in r[1]]), ('dual acc  ', [r for r in x3 if r[1].endswith('d')])):
        tx, nx, ex = min(sel)
        print('      best %s %-16s %7.1f us %6.1f TF (x%.2f) err max %.2e rms %.2e | %s'
              % (kind, nx, tx * 1e3, fl / tx / 1e9, tb / tx, ex[0], ex[1], ' '.join('%s=%.0f' % (n.split(':')[1], t * 1e3) for t, n, _ in sel)), flush=True)
    be.enable_x3(st, None)
