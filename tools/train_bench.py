#!/usr/bin/env python3
"""One-GPU training-step timing: RFBNet forward (batch-stat BN) + MultiBoxLoss + HIP backward + SGD.
   python tools/train_bench.py [--size 300 --batch 32 --classes 20 --steps 5]"""
import argparse, os, sys, time, types
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'context-transformer_amd')); sys.path.insert(0, REPO)
ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=300); ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--classes', type=int, default=20); ap.add_argument('--steps', type=int, default=5)
ap.add_argument('--sync', type=int, default=0)
ap.add_argument('--phase', type=int, default=1); ap.add_argument('--setting', default='transfer')
a = ap.parse_args()
from ctdet import synth, dist as cdist
from models.RFB_Net_vgg import build_net
from layers.functions import PriorBox
from layers.modules.multibox_loss_combined import MultiBoxLoss_combined
import data as cfgs
rank, local, world = cdist.init('nccl')
torch.cuda.set_device(local)
net = build_net(types.SimpleNamespace(method='ours', phase=a.phase, setting=a.setting), a.size, a.classes)
net.load_state_dict(synth.fill_state_dict(net.state_dict()))
net = net.cuda().train(); net.device = 'cuda'
priors = PriorBox(getattr(cfgs, 'VOC_%d' % a.size)).forward().cuda()
nout = a.classes if a.phase == 1 else net.OBJ_Target.weight.shape[0] + (a.classes if a.setting == 'incre' else 0)
crit = MultiBoxLoss_combined(nout + 1, 0.5, True, 0, True, 3, 0.5, False)
crit.sync_normalizer = world > 1
opt = torch.optim.SGD(net.parameters(), lr=1e-4, momentum=0.9, weight_decay=5e-4)
x = synth.images(a.batch, a.size, 'randn', 1234 + rank).cuda()
tg = [t.cuda() for t in synth.targets(a.batch, nout + 1, 99 + rank)]
trt = net.train_runtime(a.batch)
if a.sync or world > 1:
    trt.enable_grad_sync()
def step():
    opt.zero_grad(set_to_none=True)
    out = net(x)
    ld = crit(out, priors, tg)
    loss = sum(ld.values())
    loss.backward()
    opt.step()
    return loss
for _ in range(2):
    l = step()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
tf = tl = tb = to = 0.0
cdist.barrier('cuda')
t0 = time.perf_counter()
for _ in range(a.steps):
    opt.zero_grad(set_to_none=True)
    ev[0].record(); out = net(x); ev[1].record()
    ld = crit(out, priors, tg); loss = sum(ld.values()); ev[2].record()
    loss.backward(); ev[3].record()
    opt.step()
    torch.cuda.synchronize()
    tf += ev[0].elapsed_time(ev[1]); tl += ev[1].elapsed_time(ev[2]); tb += ev[2].elapsed_time(ev[3])
cdist.barrier('cuda')
dt = cdist.max_over_ranks((time.perf_counter() - t0) / a.steps, 'cuda')
if rank == 0:
    n = a.steps
    print('RFBNet-%d phase %d bs=%d x %d GPU(s): %.1f ms/step = %.1f img/s | fwd %.1f  loss %.1f  bwd %.1f ms | loss %.4f'
          % (a.size, a.phase, a.batch, world, dt * 1e3, a.batch * world / dt, tf / n, tl / n, tb / n, float(loss)))
