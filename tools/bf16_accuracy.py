"""bf16 channels-last mode against the fp32 path on the same weights and images: max |difference| of the raw head
outputs relative to their range (what DESIGN.md quotes for BASELINE configs[4])."""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'context-transformer_amd'))
from ctdet import synth  # noqa: E402
from models.RFB_Net_vgg import build_net  # noqa: E402

for size, kind in ((300, 'randn'), (300, 'image'), (512, 'randn')):
    net = build_net(types.SimpleNamespace(method='ours', phase=1, setting='transfer'), size, 20)
    net.load_state_dict(synth.fill_state_dict(net.state_dict()))
    net = net.cuda().eval()
    net.device = 'cuda'
    x = synth.images(4, size, kind, 4321).cuda()
    with torch.no_grad():
        ref = [t.clone() for t in net.forward_raw(x)]
        net.conv_dtype = 'bf16'
        got = net.forward_raw(x)
    torch.cuda.synchronize()
    errs = ['%s %.2e (rms %.2e)' % (n, (a - b).abs().max().item() / b.abs().max().item(),
                                   ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item())
            for n, a, b in zip(('loc', 'conf', 'obj'), got, ref)]
    print('RFBNet-%d %s images: ' % (size, kind) + ', '.join(errs))
