#!/usr/bin/env python3
"""Time the Winograd kernel against every direct tile config on the 3x3 s1 d1 layers of RFBNet-300/512 (bs 32)."""
import os, sys, types
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'context-transformer_amd')); sys.path.insert(0, REPO)
from ctdet import engine, synth
from models.RFB_Net_vgg import build_net
size = int(sys.argv[1]) if len(sys.argv) > 1 else 300
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
os.environ['CTDET_TUNE'] = '0'
net = build_net(types.SimpleNamespace(method='ours', phase=1, setting='transfer'), size, 20)
net.load_state_dict(synth.fill_state_dict(net.state_dict()))
net = net.eval().cuda(); net.device = 'cuda'
rt = net.runtime(batch)
rt.bufs['x'].normal_()
rt.run_backbone(rt.bufs['x'].clone())
be = rt.backend
tot_d = tot_w = 0.0
for st in rt.conv_steps():
    if not st.rt.get('wino_ok'):
        continue
    best, times = be.tune_conv(st, iters=5)
    d, w = min(times[:-1]), times[-1]
    fl = st.flops(batch)
    tot_d += d; tot_w += min(d, w)
    print('%-18s %4d->%-4d @%3dx%-3d direct %8.1f us (%5.1f TF)  wino %8.1f us (%5.1f TF eff)  x%.2f %s'
          % (st.name, st.cin, st.cout, st.h, st.w, d * 1e3, fl / d / 1e9, w * 1e3, fl / w / 1e9, d / w, 'WINO' if st.rt.get('wino') else ''))
print('sum direct %.2f ms -> with wino %.2f ms' % (tot_d, tot_w))
