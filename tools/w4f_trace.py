#!/usr/bin/env python3
"""Where a workgroup of the fused F(4x4,3x3) / bf16x3 kernel (csrc/ct_wino4f.hip) spends its life: shader-clock stamps of
wave 0 (s_memrealtime, 100 MHz) at the phase boundaries, from a measurement build of the library:
    CTDET_EXTRA_FLAGS=-DCTDET_W4F_TRACE python context-transformer_amd/build.py --force && python tools/w4f_trace.py [shape ...]
Columns: prologue (entry -> V(0) complete), main loop, first / second output pass, and the gap between consecutive
workgroups on a CU slot (end of one -> entry of the next one that started after it, estimated from the sorted stamps)."""
import ctypes as C, os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'context-transformer_amd')); sys.path.insert(0, REPO)
from ctdet import _lib, engine
DEV = 'cuda:0'
SHAPES = {'base.2': (32, 64, 300, 300, 64), 'base.5': (32, 64, 150, 150, 128), 'base.7': (32, 128, 150, 150, 128),
          'base.10': (32, 128, 75, 75, 256), 'base.12': (32, 256, 75, 75, 256)}
lib = _lib.lib()
lib.ct_wino4f_set_trace.argtypes = [C.c_void_p]
be = engine.HipBackend(DEV)
for name in sys.argv[1:] or ['base.2', 'base.7']:
    B, Cin, H, W, Cout = SHAPES[name]
    w = torch.nn.Parameter(torch.randn(Cout, Cin, 3, 3, device=DEV) * 0.05, requires_grad=False)
    b = torch.nn.Parameter(torch.zeros(Cout, device=DEV), requires_grad=False)
    st = engine.ConvStep(name, [engine.ConvPart(w, b, None, True)], Cin, 3, 3, 1, 1, 1, 1, 'x', 0, H, W, 'y', 0)
    bufs = {'x': torch.randn(B, Cin, H, W, device=DEV), 'y': torch.empty(B, Cout, H, W, device=DEV)}
    be.prepare_conv(st, bufs, B)
    tile = int(os.environ.get('TILE', '46'))            # 46: bf16x3, 48: f16x2 (needs the maximum of |input|, taken once here)
    be.enable_wino(st, tile=tile)
    if tile == 48:
        slot = torch.zeros(B * _lib.ABSMAX_LINE_BYTES // 4, device=DEV, dtype=torch.int32)
        _lib.check(be.lib.ct_absmax_f32(bufs['x'].data_ptr(), B, Cin * H * W, Cin * H * W, slot.data_ptr(), be._stream()), 'ct_absmax_f32')
        st.rt['desc'].in_absmax = slot.data_ptr()
    for _ in range(3):
        be.run_conv(st)
    tiles = B * ((H + 3) // 4) * ((W + 3) // 4)
    nwg = min(256, 8 * (((tiles + 31) // 32 + 7) // 8) * ((Cout + 63) // 64))      # persistent grid: the stamps of a workgroup's LAST item remain
    trace = torch.zeros(nwg * 8, dtype=torch.int64, device=DEV)
    _lib.check(lib.ct_wino4f_set_trace(trace.data_ptr()), 'set_trace')
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); be.run_conv(st); e1.record(); torch.cuda.synchronize()
    _lib.check(lib.ct_wino4f_set_trace(None), 'set_trace')
    t = trace.view(nwg, 8).cpu().double()
    n_all = int((t[:, 4] > 0).sum())
    t = t[(t[:, :5] > 0).all(1)]
    if len(t) < n_all:
        print('   (%d of %d workgroups have an incomplete set of stamps: dropped)' % (n_all - len(t), n_all))
    us = e0.elapsed_time(e1) * 1e3
    span = float(t[:, 4].max() - t[:, 0].min())
    tick = 0.01                # s_memrealtime: 100 MHz
    d = [(t[:, k + 1] - t[:, k]) * tick for k in range(4)]
    print('%-8s %d->%d @%dx%d bs%d: launch %.1f us, %d workgroups (%.2f rounds of 256), first entry -> last exit %.1f us by the 100 MHz counter'
          % (name, Cin, Cout, H, W, B, us, len(t), len(t) / 256.0, span * tick))
    for lab, v in zip(('first phase T', 'main loop', 'output q0+q1', 'output q2+q3'), d):
        print('   %-14s mean %6.2f us   min %6.2f   max %6.2f' % (lab, v.mean(), v.min(), v.max()))
    life = (t[:, 4] - t[:, 0]) * tick
    print('   %-14s mean %6.2f us per item (last item of each of the %d persistent workgroups)' % ('whole item', life.mean(), len(t)))
