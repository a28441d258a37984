#!/usr/bin/env python3
"""Time the batched post-processing stages on realistic model outputs (debug aid)."""
import os, sys, types
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'context-transformer_amd')); sys.path.insert(0, REPO)
os.environ.setdefault('CTDET_TUNE', '0')
from ctdet import ops, synth
from ctdet.pipeline import DetectionPipeline
from models.RFB_Net_vgg import build_net
from layers.functions import PriorBox
import data as cfgs

B = int(os.environ.get('B', 32))
SIZE = int(os.environ.get('SIZE', 300))          # SIZE=512: RFBNet-512's 32 756 priors
net = build_net(types.SimpleNamespace(method='ours', phase=1, setting='transfer'), SIZE, 20)
net.load_state_dict(synth.fill_state_dict(net.state_dict()))
net = net.eval().cuda(); net.device = 'cuda'
pipe = DetectionPipeline(net, PriorBox(getattr(cfgs, 'VOC_%d' % SIZE)).forward(), B, 20)
x = synth.images(B, SIZE, 'randn', 1234).cuda()
pipe.run(x)
torch.cuda.synchronize()
cnt = pipe.post.ws  # noqa
sc = pipe.scores
ncand = (sc[:, :, 1:] > 0.01).sum(1).float()
print('candidates per (img,cls): mean %.0f max %d' % (ncand.mean().item(), int(ncand.max().item())))
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print('postprocess (select+sort+nms+topk+gather): %.3f ms' % t(lambda: pipe.post.run(pipe.boxes, pipe.scores)))
def stage_times(n=5):
    import ctypes as C
    from ctdet import _lib
    lib = _lib.lib()
    torch.cuda.synchronize()
    _lib.check(lib.ct_profile_enable(1), 'ct_profile_enable')
    for _ in range(n):
        pipe.post.run(pipe.boxes, pipe.scores)
    torch.cuda.synchronize()
    cnt = C.c_int(0)
    _lib.check(lib.ct_profile_collect(None, 0, C.byref(cnt)), 'ct_profile_collect')
    recs = (_lib.ProfileRecord * max(cnt.value, 1))()
    _lib.check(lib.ct_profile_collect(recs, cnt.value, C.byref(cnt)), 'ct_profile_collect')
    _lib.check(lib.ct_profile_enable(0), 'ct_profile_enable')
    agg = {}
    for i in range(cnt.value):
        a = agg.setdefault(recs[i].name.decode(), [0.0, 0])
        a[0] += recs[i].ms; a[1] += 1
    return {k: (v[0] / v[1] * 1e3, v[1] / n) for k, v in agg.items()}
print('  stages (us per launch x launches per run): ' + '  '.join('%s %.1f x%g' % (k, us, c) for k, (us, c) in sorted(stage_times().items())))
print('postprocess without top-k rule:            %.3f ms' % t(lambda: pipe.post.run(pipe.boxes, pipe.scores, max_per_image=0)))
pipe.post.run(pipe.boxes, pipe.scores, max_per_image=0)
print('kept per (img,cls) before top-k: mean %.0f max %d' % (pipe.post.out_count.float().mean().item(), int(pipe.post.out_count.max().item())))
loc, conf, obj = net.forward_raw(x)
print('detect_fused: %.3f ms' % t(lambda: ops.detect_fused(loc, conf.contiguous(), obj, pipe.priors, (0.1, 0.2), True, pipe.scale, out=(pipe.boxes, pipe.scores))))

# ---- SURVEY 8(d) regimes on synthetic detections (no model): R1 "all-pass" = every prior of every class is a
# candidate (640 problems of N = P); R2 "trained-like" = 50..400 clustered boxes per (image, class), tie-free scores
P, T = pipe.P, 20
g = torch.Generator().manual_seed(4321)
pri = pipe.priors.cpu()
boxes = torch.cat([pri[:, :2] - pri[:, 2:] / 2, pri[:, :2] + pri[:, 2:] / 2], 1).clamp(0, 1)
boxes = (boxes[None].repeat(B, 1, 1) * torch.tensor([500., 375., 500., 375.])).contiguous().cuda()
scores = torch.zeros(B, P, T + 1)
scores[:, :, 1:] = 0.011 + 0.97 * torch.rand(B, P, T, generator=g)
scores = scores.cuda()
for topk in (200, 0):
    ms = t(lambda: pipe.post.run(boxes, scores, max_per_image=topk), 5)
    print('R1 all-pass (N = P = %d per class, %d problems), max_per_image=%d: %.2f ms' % (P, B * T, topk, ms))
scores2 = torch.zeros(B, P, T + 1)
for b in range(B):
    for c in range(T):
        n = int(torch.randint(50, 401, (1,), generator=g))
        idx = torch.randperm(P, generator=g)[:n]
        scores2[b, idx, 1 + c] = torch.linspace(0.011, 0.99, n)[torch.randperm(n, generator=g)]
scores2 = scores2.cuda()
for topk in (200, 0):
    ms = t(lambda: pipe.post.run(boxes, scores2, max_per_image=topk), 10)
    print('R2 trained-like (50..400 per class), max_per_image=%d: %.3f ms' % (topk, ms))

# the reference's own call protocol (test.py:152): one host call per (image, class) through the `_nms` contract
from utils.nms_wrapper import nms
import time
for n in (50, 300, 2000):
    d = synth.clustered_dets(n)
    nms(d, 0.45)
    t0 = time.perf_counter()
    for _ in range(50):
        nms(d, 0.45)
    print('utils.nms_wrapper.nms (host `_nms` contract: H2D + kernel + D2H), %4d boxes: %.1f us per call' % (n, (time.perf_counter() - t0) / 50 * 1e6))
