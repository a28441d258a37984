import os, sys, types
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'context-transformer_amd')); sys.path.insert(0, REPO)
from ctdet import synth
from oracle import rfbnet_ref
from models.RFB_Net_vgg import build_net
net = build_net(types.SimpleNamespace(method='ours', phase=1, setting='transfer'), 300, 20)
net.load_state_dict(synth.fill_state_dict(net.state_dict()))
net = net.cuda().train(); net.device = 'cuda'
x = synth.images(2, 300, 'randn', 1234)
sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
g = torch.Generator().manual_seed(0)
out = net(x.cuda())
rs = [torch.randn(o.shape, generator=g) for o in out]
loss = sum((o * r.cuda()).sum() for o, r in zip(out, rs))
loss.backward()
cap = {}
orig = rfbnet_ref.backbone
def capture(*a, **k):
    srcs = orig(*a, **k)
    for s in srcs: s.retain_grad()
    cap['srcs'] = srcs
    return srcs
rfbnet_ref.backbone = capture
leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and 'running' not in k}
sdo = dict(sd); sdo.update(leaf)
oo = rfbnet_ref.forward(sdo, x, 300, 20, training=True)
sum((o * r).sum() for o, r in zip(oo, rs)).backward()
trt = net.train_runtime(2)
names = ['Norm.out', 'extras.0.out', 'extras.1.out', 'extras.2.out', 'a_extras.4', 'a_extras.6']
for n, s in zip(names, cap['srcs']):
    a, b = trt.grads[n].cpu(), s.grad
    fa, fb = trt.bufs[n].cpu(), s.detach()
    print('%-14s fwd rel %.2e | grad d=%.3e |g|=%.3e rel=%.2e' % (n, float((fa-fb).abs().max()/fb.abs().max()),
          float((a-b).abs().max()), float(b.abs().max()), float((a-b).abs().max()/b.abs().max())))
    if n == 'Norm.out':
        d = (a-b).abs()
        idx = torch.nonzero(d > 0.01 * b.abs().max())
        print('  n bad', idx.shape[0], 'of', d.numel(), 'first', idx[:8].tolist())
        print('  per-channel max err (first 16 ch):', [round(float(d[:, c].max()), 3) for c in range(16)])
        print('  border vs interior: ', float(d[:, :, 1:-1, 1:-1].max()), float(d.max()))
