import sys, os, types
sys.path.insert(0,'/root/repo/context-transformer_amd'); sys.path.insert(0,'/root/repo')
import torch
from ctdet import synth
from models.RFB_Net_vgg import build_net
net = build_net(types.SimpleNamespace(method='ours', phase=1, setting='transfer'), 300, 20)
net.load_state_dict(synth.fill_state_dict(net.state_dict())); net = net.cuda().train(); net.device='cuda'
trt = net.train_runtime(32)
for st in trt.plan.steps:
    if st.kind=='conv':
        s = trt.state[st.name]
        if s.fwd.rt.get('wino') or getattr(s,'dgrad_tile',None):
            print(st.name, st.cin, st.cout, st.h, 'fwd', s.fwd.rt.get('wino'), 'dgrad', getattr(s,'dgrad_tile',None) if s.dgrad is not None and s.dgrad_wino is not None else None, 'wgrad', s.wgrad_tile if s.wgrad_wino else None)
