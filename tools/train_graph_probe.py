#!/usr/bin/env python3
"""Which part of a training step survives torch.cuda.graph capture on this ROCm (one part per process: a failure is a
segfault inside hipStreamEndCapture).   python tools/train_graph_probe.py [part]   part in fwd / loss / bwd / opt / all"""
import os, subprocess, sys, time, types
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) < 2:
    for part in ('fwd', 'loss', 'lossonly', 'bwd', 'opt'):
        r = subprocess.run([sys.executable, '-X', 'faulthandler', __file__, part], capture_output=True, text=True, timeout=600)
        tail = [l for l in (r.stdout + r.stderr).splitlines() if l.startswith('RESULT') or l.startswith('SIZES') or 'Error' in l or 'error' in l][-3:]
        print('%-9s rc=%d  %s' % (part, r.returncode, ' | '.join(tail)[:300]), flush=True)
    sys.exit(0)
part = sys.argv[1]
import torch
sys.path.insert(0, os.path.join(REPO, 'context-transformer_amd')); sys.path.insert(0, REPO)
from ctdet import synth
from models.RFB_Net_vgg import build_net
from layers.functions import PriorBox
from layers.modules.multibox_loss_combined import MultiBoxLoss_combined
import data as cfgs
B = int(os.environ.get('B', 8))
net = build_net(types.SimpleNamespace(method='ours', phase=1, setting='transfer'), 300, 20)
net.load_state_dict(synth.fill_state_dict(net.state_dict()))
net = net.cuda().train(); net.device = 'cuda'
priors = PriorBox(cfgs.VOC_300).forward().cuda()
crit = MultiBoxLoss_combined(21, 0.5, True, 0, True, 3, 0.5, False)
opt = torch.optim.SGD(net.parameters(), lr=1e-4, momentum=0.9, weight_decay=5e-4)
x = synth.images(B, 300, 'randn', 1234).cuda()
tg = [t.cuda() for t in synth.targets(B, 21, 99)]
trt = net.train_runtime(B)
matched = crit.match(priors, tg, 'cuda')
print('SIZES arena %.1f MB, bn_scratch %.1f MB, wgrad_ws_all %.1f MB' % (trt.arena.numel() * 4 / 2**20, trt.bn_scratch.numel() * 4 / 2**20, trt.wgrad_ws_all.numel() * trt.wgrad_ws_all.element_size() / 2**20), flush=True)
static_out = None


def region():
    global static_out
    if part == 'lossonly':
        return sum(crit(static_out, priors, matched).values())
    out = net(x)
    if part == 'fwd':
        return out[0].sum()
    loss = sum(crit(out, priors, matched).values())
    if part == 'loss':
        return loss
    loss.backward()
    if part == 'opt':
        opt.step()
    return loss


if part == 'lossonly':
    with torch.no_grad():
        static_out = tuple(t.detach().clone() for t in net(x))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        region()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    val = region()
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
print('RESULT %s captured, replay %.2f ms, value %.4f' % (part, (time.perf_counter() - t0) / 5 * 1e3, float(val)))
