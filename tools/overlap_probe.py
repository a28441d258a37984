#!/usr/bin/env python3
"""Do two detection steps in flight (two pipelines, two streams, alternating batches) beat one after the other?
   python tools/overlap_probe.py [--batch 32 --steps 20]"""
import argparse, faulthandler, os, sys, time, types
import torch
faulthandler.enable()
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'context-transformer_amd')); sys.path.insert(0, REPO)
from ctdet import synth
from ctdet.pipeline import DetectionPipeline
from models.RFB_Net_vgg import build_net
from layers.functions import PriorBox
import data as cfgs
ap = argparse.ArgumentParser(); ap.add_argument('--batch', type=int, default=32); ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--n', type=int, default=2)
a = ap.parse_args()
priors = PriorBox(cfgs.VOC_300).forward()
pipes, xs, streams = [], [], []
for i in range(a.n):
    net = build_net(types.SimpleNamespace(method='ours', phase=1, setting='transfer'), 300, 20)
    net.load_state_dict(synth.fill_state_dict(net.state_dict()))
    net = net.eval().cuda(); net.device = 'cuda'
    pipes.append(DetectionPipeline(net, priors, a.batch, 20))
    xs.append(synth.images(a.batch, 300, 'randn', 1234 + i).cuda())
    streams.append(torch.cuda.Stream())
# every pipeline is captured and replayed on ITS OWN stream (replaying a hipGraph on another stream than the one it was
# captured on segfaults in hipGraphLaunch on ROCm 7.2)
for p, x, s in zip(pipes, xs, streams):
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(5):
            p.run(x)
torch.cuda.synchronize()
print('warm', flush=True)
t0 = time.perf_counter()
with torch.cuda.stream(streams[0]):
    for k in range(a.steps):
        pipes[0].run(xs[0])
torch.cuda.synchronize()
t1 = (time.perf_counter() - t0) / a.steps
print('single done', flush=True)
t0 = time.perf_counter()
for k in range(a.steps):
    i = k % a.n
    with torch.cuda.stream(streams[i]):
        pipes[i].run(xs[i])
torch.cuda.synchronize()
t2 = (time.perf_counter() - t0) / a.steps
print('bs %d: one pipeline %.3f ms/step (%.0f img/s); %d pipelines in flight %.3f ms/step (%.0f img/s)'
      % (a.batch, t1 * 1e3, a.batch / t1, a.n, t2 * 1e3, a.batch / t2))
