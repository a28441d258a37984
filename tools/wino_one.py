#!/usr/bin/env python3
"""One Winograd layer, N launches (for rocprofv3 --pmc passes and A/B timing):
   python tools/wino_one.py [shape ...]    shapes: base.2 base.7 base.12 base.19 base.24 head.0"""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'context-transformer_amd')); sys.path.insert(0, REPO)
from ctdet import _lib, engine
DEV = 'cuda:0'
SHAPES = {  # B, Cin, H, W, Cout
    'base.2': (32, 64, 300, 300, 64), 'base.5': (32, 64, 150, 150, 128), 'base.7': (32, 128, 150, 150, 128),
    'base.10': (32, 128, 75, 75, 256), 'base.12': (32, 256, 75, 75, 256), 'base.17': (32, 256, 38, 38, 512),
    'base.19': (32, 512, 38, 38, 512), 'base.24': (32, 512, 19, 19, 512), 'head.0': (32, 512, 38, 38, 156),
    'base.19.b4': (4, 512, 38, 38, 512), 'base.2.b4': (4, 64, 300, 300, 64),
    'base.24.b4': (4, 512, 19, 19, 512), 'head.0.b4': (4, 512, 38, 38, 156), 'head.1.b4': (4, 1024, 19, 19, 156),
    'base.24.b8': (8, 512, 19, 19, 512), 'base.14': (32, 256, 75, 75, 256), 'head.1': (32, 1024, 19, 19, 156),
    'base.17b': (32, 256, 38, 38, 512), 'c512.4': (32, 512, 64, 64, 512),
    'base.7.b2': (2, 128, 150, 150, 128), 'base.7.b4': (4, 128, 150, 150, 128), 'base.7.b8': (8, 128, 150, 150, 128), 'base.10.b8': (8, 128, 75, 75, 256),
    'base.2.b1': (1, 64, 300, 300, 64), 'base.2.b2b': (2, 64, 300, 300, 64), 'base.5.b4': (4, 64, 150, 150, 128), 'base.5.b8': (8, 64, 150, 150, 128),
    'small': (2, 16, 21, 37, 40), 'tiny': (1, 8, 8, 8, 64), 'b2.b2': (2, 64, 300, 300, 64),
}
names = sys.argv[1:] or ['base.2', 'base.7', 'base.12', 'base.19', 'base.24']
iters = int(os.environ.get('ITERS', 10))
be = engine.HipBackend(DEV)
for name in names:
    B, Cin, H, W, Cout = SHAPES[name]
    w = torch.nn.Parameter(torch.randn(Cout, Cin, 3, 3, device=DEV) * 0.05, requires_grad=False)
    b = torch.nn.Parameter(torch.zeros(Cout, device=DEV), requires_grad=False)
    st = engine.ConvStep(name, [engine.ConvPart(w, b, None, True)], Cin, 3, 3, 1, 1, 1, 1, 'x', 0, H, W, 'y', 0)
    bufs = {'x': torch.randn(B, Cin, H, W, device=DEV), 'y': torch.empty(B, Cout, H, W, device=DEV)}
    if os.environ.get('ZERO'):                # all-zero activations: same instruction stream, idle multipliers (power experiment)
        bufs['x'].zero_()
    be.prepare_conv(st, bufs, B)
    if os.environ.get('SK'):                  # stream-K: the launch has the device to itself
        skws = torch.empty(256 * 2 * 64 * 32 * 16, device=DEV)
        d = st.rt['desc']
        d.ksplit, d.ksplit_ws, d.ksplit_ws_floats = -2, skws.data_ptr(), skws.numel()
    ref = None
    if os.environ.get('CHECK', '1') != '0' and B * Cin * H * W <= 64 << 20:
        ref = torch.relu(torch.nn.functional.conv2d(bufs['x'].double(), w.double(), b.double(), padding=1))
    for tile in [int(t) for t in os.environ.get('TILES', '2,4,23,44,46,47,48').split(',')]:
        if tile in (23, 44, 46, 47, 48) and Cin % 16:
            continue
        be.enable_wino(st, tile=tile)
        st.rt['desc'].in_absmax = None
        st.rt.pop('amax_own', None)
        if tile in (47, 48) and os.environ.get('PRE_AMAX', '1') != '0':
            # the f16x2 kernels take their input scale from a maximum of |input|: in a network the producer of the input leaves it
            # (ct_conv_desc.out_absmax -> in_absmax); here it is taken ONCE, outside the timed launches (PRE_AMAX=0: inside them)
            slot = torch.zeros(B * _lib.ABSMAX_LINE_BYTES // 4, device=DEV, dtype=torch.int32)
            _lib.check(be.lib.ct_absmax_f32(bufs['x'].data_ptr(), B, Cin * H * W, Cin * H * W, slot.data_ptr(), be._stream()), 'ct_absmax_f32')
            st.rt['desc'].in_absmax = slot.data_ptr()
            st.rt['slot_keepalive'] = slot
        bufs['y'].fill_(float('nan'))
        for _ in range(2):
            be.run_conv(st)
        torch.cuda.synchronize()
        err = ''
        if ref is not None:
            err = '  max err %.2e of range' % ((bufs['y'].double() - ref).abs().max() / ref.abs().max()).item()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            be.run_conv(st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        fl = st.flops(B)
        # executed matrix work: F(2x2) 16/36, F(4x4) 36/144 of the direct count; bf16x3 = six bf16 products each
        # f16x2 (47, 48): three f16 products each
        ex = 0.25 if tile == 4 else 0.25 * 6 if tile in (44, 46) else 0.25 * 3 if tile in (47, 48) else (16 / 36) * (6 if tile == 23 else 1)
        peak = 2500.0 if tile in (23, 44, 46, 47, 48) else 157.3
        stages = ''
        if os.environ.get('STAGES') and tile in (44, 47):
            # the three kernels of the launch by the library's own profile scopes (HIP events on the launch stream)
            import ctypes as C
            lib = _lib.lib()
            _lib.check(lib.ct_profile_enable(1), 'ct_profile_enable')
            for _ in range(iters):
                be.run_conv(st)
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
            stages = '  [' + '  '.join('%s %.1f us' % (k.replace('wino4s_', '').replace('wino4h_', ''), v[0] / v[1] * 1e3) for k, v in agg.items()) + ']'
        print('%-10s F%-2d %4d->%-4d @%3dx%-3d bs%-2d  %8.1f us  %6.1f TF algorithmic  %6.1f TF executed = %.3f of %.1f%s'
              % (name, tile, Cin, Cout, H, W, B, ms * 1e3, fl / ms / 1e9, fl / ms / 1e9 * ex, fl / ms / 1e9 * ex / peak, peak, err + stages),
              flush=True)
