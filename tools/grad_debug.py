#!/usr/bin/env python3
"""Per-parameter gradient error table of one RFBNet training step (HIP backward vs torch-CPU fp32 vs float64 autograd
through the oracle; a random linear loss on the raw head outputs).  Study tool behind the tolerances of
tests/test_gpu_train.py / test_gpu_dp_train.py:  python tools/grad_debug.py [--batch 8] [--frozen-bn] [--size 300]"""
import argparse, os, sys, types
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'context-transformer_amd')); sys.path.insert(0, REPO)
from ctdet import synth
from oracle import rfbnet_ref
from models.RFB_Net_vgg import build_net
ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=2)
ap.add_argument('--size', type=int, default=300)
ap.add_argument('--frozen-bn', action='store_true')
ap.add_argument('--seed', type=int, default=1234)
a = ap.parse_args()
net = build_net(types.SimpleNamespace(method='ours', phase=1, setting='transfer'), a.size, 20)
net.load_state_dict(synth.fill_state_dict(net.state_dict()))
net = net.cuda().train(); net.device = 'cuda'
if a.frozen_bn:
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
x = synth.images(a.batch, a.size, 'randn', a.seed)
sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
g = torch.Generator().manual_seed(0)
out = net(x.cuda())
rs = [torch.randn(o.shape, generator=g) / o.numel() ** 0.5 for o in out]
loss = sum((o * r.cuda()).sum() for o, r in zip(out, rs))
loss.backward()
kw = dict(raw=True) if a.frozen_bn else dict(training=True)
leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and 'running' not in k}
sdo = dict(sd); sdo.update(leaf)
oo = rfbnet_ref.forward(sdo, x, a.size, 20, **kw)
lo = sum((o * r).sum() for o, r in zip(oo, rs))
lo.backward()
leaf64 = {k: v.double().clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and 'running' not in k}
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}; sd64.update(leaf64)
o64 = rfbnet_ref.forward(sd64, x.double(), a.size, 20, **kw)
l64 = sum((o * r.double()).sum() for o, r in zip(o64, rs))
l64.backward()
print('loss', loss.item(), lo.item(), l64.item())
for o, p, q, n in zip(out, oo, o64, ('loc', 'conf', 'obj')):
    s = float(q.abs().max())
    print('%-5s fwd gpu_vs_f64=%.2e cpu32_vs_f64=%.2e' % (n, float((o.detach().cpu().double() - q).abs().max()) / s,
                                                        float((p.detach().double() - q).abs().max()) / s))
for name, prm in net.named_parameters():
    aa, b, c = prm.grad.cpu().double(), leaf[name].grad.double(), leaf64[name].grad
    nb = float(c.abs().max())
    print('%-40s |g|=%.3e gpu_vs_f64=%.2e cpu32_vs_f64=%.2e' % (name, nb, float((aa - c).abs().max()) / (nb + 1e-30),
                                                                 float((b - c).abs().max()) / (nb + 1e-30)))
