#!/usr/bin/env python3
"""Whole-network rounding error of the conv variants against an fp64 evaluation of the same network
(oracle/rfbnet_ref.py in double precision -- measurement tool, not product):
    python tools/wino_accuracy.py [--size 300] [--batch 2]
Every 3x3 / stride 1 / dilation 1 layer the committed tune table routes through a Winograd kernel is forced to one variant
(F(2x2,3x3) / F(4x4,3x3) on the fp32 MFMA, F(2x2,3x3) / three-kernel F(4x4,3x3) / fused F(4x4,3x3) on bf16x3); the table gives the largest error of loc / conf / obj over the output range, for phase 1 and for phase 2
(after the Context-Transformer block), against fp64 and against the reference's fp32 CPU arithmetic."""
import argparse, os, sys, types
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'context-transformer_amd')); sys.path.insert(0, REPO)
os.environ['CTDET_TUNE'] = '0'
from ctdet import synth  # noqa: E402
from models.RFB_Net_vgg import build_net  # noqa: E402
from oracle import rfbnet_ref  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=300)
ap.add_argument('--batch', type=int, default=2)
a = ap.parse_args()
for phase, C, kind in ((1, 20, 'randn'), (1, 20, 'u8'), (2, 60, 'randn')):
    net = build_net(types.SimpleNamespace(method='ours', phase=phase, setting='transfer'), a.size, C)
    net.load_state_dict(synth.fill_state_dict(net.state_dict()))
    net = net.eval().cuda(); net.device = 'cuda'
    x = synth.images(a.batch, a.size, kind, 1234)
    sd64 = {k: (v.detach().cpu().double() if v.is_floating_point() else v.detach().cpu()) for k, v in net.state_dict().items()}
    sd32 = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    with torch.no_grad():
        want = rfbnet_ref.forward(sd64, x.double(), a.size, C, phase, 'ours', 'transfer', raw=True)
        ref32 = rfbnet_ref.forward(sd32, x, a.size, C, phase, 'ours', 'transfer', raw=True)

    def report(name, n, got):
        e64 = ['%s %.2e' % (nm, float((g.reshape(w.shape) - w).abs().max() / w.abs().max()))
               for nm, g, w in zip(('loc', 'conf', 'obj'), got, want)]
        e32 = ['%.2e' % float((g.reshape(w.shape) - w.double()).abs().max() / w.abs().max()) for g, w in zip(got, ref32)]
        print('RFBNet-%d phase %d %-5s input, %2d layers on %-18s vs fp64: %s   vs the fp32 CPU path: %s'
              % (a.size, phase, kind, n, name, '  '.join(e64), ' '.join(e32)), flush=True)

    report('(the fp32 CPU path itself)', 0, [t.double() for t in ref32])
    rt = net.runtime(a.batch)
    with torch.no_grad():
        got = [t.double().cpu() for t in net.forward_raw(x.cuda())]
    if got is not None:          # what the runtime ships: the committed table, f16x2 operand forms where operand_form_h2 says so
        from ctdet import engine as _eng
        kinds = {}
        for st in rt.conv_steps():
            k = _eng.WINO_NAME.get(st.rt.get('wino')) or (rt.backend.x3_names()[st.rt['x3']][:2] if st.rt.get('x3') is not None else 'fp32')
            kinds[k] = kinds.get(k, 0) + 1
        report('the shipped table (CTDET_H2=%s): %s' % (os.environ.get('CTDET_H2', '1'), ' '.join('%s %d' % kv for kv in sorted(kinds.items()))),
               len(rt.conv_steps()), got)
    # the layers the committed table routes through a Winograd kernel and that have the fused kernels' geometry (3x3, stride 1,
    # dilation 1): the dilated layers only exist on the three-kernel form (tile 44) and keep it in every row
    fused = [st for st in rt.conv_steps() if st.rt.get('wino') and st.rt.get('wino_ok')]
    variants = [('F(2x2,3x3) fp32', 2), ('F(4x4,3x3) fp32', 4), ('F(2x2) bf16x3 2acc', 23), ('F(4x4) 3-kernel', 44), ('F(4x4) fused x3', 46)]
    for name, tile in variants:
        n = 0
        for st in fused:
            ok = {23: 'winox_ok', 44: 'wino4s_ok', 46: 'wino4f_ok'}.get(tile)
            rt.backend.enable_wino(st, True, tile=tile if (ok is None or st.rt.get(ok)) else 2)      # conv1_1-like layers: no 16-channel chunks
            n += 1
        with torch.no_grad():
            got = [t.double().cpu() for t in net.forward_raw(x.cuda())]
        report(name, n, got)
