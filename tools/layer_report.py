#!/usr/bin/env python3
"""Per-launch timing table of the RFBNet engine on the MI355X (HIP events on the launch stream).

    python tools/layer_report.py [--size 300 --batch 32 --classes 20 --phase 1 --tune 1]
Prints one row per conv launch: geometry, tile config, time, TFLOP/s, fraction of the fp32 MFMA peak.
"""
import argparse
import os
import sys
import types

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'context-transformer_amd'))
sys.path.insert(0, REPO)

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=300)
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--classes', type=int, default=20)
ap.add_argument('--phase', type=int, default=1)
ap.add_argument('--setting', default='transfer')
ap.add_argument('--tune', type=int, default=1)
ap.add_argument('--iters', type=int, default=10)
a = ap.parse_args()
os.environ['CTDET_TUNE'] = str(a.tune)

from ctdet import _lib, synth  # noqa: E402
from models.RFB_Net_vgg import build_net  # noqa: E402

net = build_net(types.SimpleNamespace(method='ours', phase=a.phase, setting=a.setting), a.size, a.classes)
net.load_state_dict(synth.fill_state_dict(net.state_dict()))
net = net.eval().cuda()
net.device = 'cuda'
rt = net.runtime(a.batch)
x = synth.images(a.batch, a.size, 'randn', 1234).cuda()
lib = _lib.lib()
with torch.no_grad():
    for _ in range(3):
        rt.run_backbone(x)
    rt.event_log = []
    for _ in range(a.iters):
        rt.run_backbone(x)
    torch.cuda.synchronize()
agg = {}
for st, e0, e1 in rt.event_log:
    agg.setdefault(st.name, [st, 0.0])[1] += e0.elapsed_time(e1) / a.iters
tot_t = tot_f = 0.0
print('%-22s %-26s %-9s %9s %8s %6s  %s' % ('step', 'geometry', 'cfg', 'us', 'TFLOP/s', 'frac', 'tune(ms per cfg)'))
for name, (st, ms) in agg.items():
    f = st.flops(a.batch)
    tot_t += ms
    tot_f += f
    cfg = st.rt['desc'].config
    geo = '%dx%d s%d d%d %d->%d @%dx%d' % (st.kh, st.kw, st.stride, st.dil, st.cin, st.cout, st.oh, st.ow)
    tune = ' '.join('%.3f' % t for t in st.rt.get('tune_ms', []))
    from ctdet import engine as _e
    kname = _e.WINO_NAME.get(st.rt.get('wino')) or (rt.backend.x3_names()[st.rt['x3']] if st.rt.get('x3') is not None else None)
    print('%-22s %-26s %-9s %9.1f %8.2f %6.3f  %s' % (name, geo, kname or (lib.ct_conv_config_name(cfg - 1).decode() if cfg else 'auto'),
                                                    ms * 1e3, f / ms / 1e9, f / ms / 1e9 / 157.3, tune))
print('TOTAL conv %.3f ms/step, %.2f TFLOP/s (%.1f%% of 157.3)' % (tot_t, tot_f / tot_t / 1e9, tot_f / tot_t / 1e9 / 1.573))
