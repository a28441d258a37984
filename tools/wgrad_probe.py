"""Time the direct and the Winograd weight-gradient kernels (F(3x3,2x2), F(3x3,4x4) fused, F(3x3,4x4) as transform kernels +
bf16x3 GEMM) on the RFBNet 3x3 layer shapes (batch 32); differences against the direct kernel."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'context-transformer_amd'))
from ctdet import _lib  # noqa: E402

SHAPES = [(int(t.split(':')[0]), int(t.split(':')[1]), int(t.split(':')[2])) for t in os.environ['WG_SHAPES'].split(',')] if os.environ.get('WG_SHAPES') else [(64, 64, 300), (64, 128, 150), (128, 128, 150), (128, 256, 75), (256, 256, 75), (256, 512, 38),
          (512, 512, 38), (512, 512, 19), (1024, 256, 19), (128, 192, 38), (256, 256, 10),
          (64, 96, 5), (128, 256, 3), (256, 24, 3), (256, 126, 1), (1024, 126, 19), (512, 126, 38), (512, 24, 38)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--iters', type=int, default=10)
    a = ap.parse_args()
    lib = _lib.lib()
    dev = 'cuda:0'
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for cin, cout, hw in SHAPES:
        x = torch.randn(a.batch, cin, hw, hw, device=dev)
        dz = torch.randn(a.batch, cout, hw, hw, device=dev)
        d = _lib.ConvDesc()
        d.in_ = x.data_ptr()
        d.batch, d.cin, d.h, d.w, d.in_ctot, d.in_coff = a.batch, cin, hw, hw, cin, 0
        d.cout, d.kh, d.kw, d.stride, d.pad_h, d.pad_w, d.dil, d.oh, d.ow = cout, 3, 3, 1, 1, 1, 1, hw, hw
        dw = torch.empty(cout, cin, 3, 3, device=dev)
        dw2 = torch.empty_like(dw)
        dw4 = torch.empty_like(dw)
        dw4s = torch.empty_like(dw)
        has4s = cin % 16 == 0 and bool(lib.ct_conv_wgrad_wino4s_supported(C.byref(d)))
        ws4s = torch.empty(lib.ct_conv_wgrad_wino4s_workspace_bytes(C.byref(d)) if has4s else 1, device=dev, dtype=torch.uint8)
        ws = torch.empty(lib.ct_conv_wgrad_wino4_workspace_bytes(C.byref(d)) // 4, device=dev)
        res = []
        for fn in ('direct', 'wino', 'wino4') + (('wino4s',) if has4s else ()):
            def run():
                if fn == 'direct':
                    _lib.check(lib.ct_conv2d_wgrad(C.byref(d), dz.data_ptr(), cout, 0, dw.data_ptr(), st), fn)
                elif fn == 'wino':
                    _lib.check(lib.ct_conv2d_wgrad_wino(C.byref(d), dz.data_ptr(), cout, 0, dw2.data_ptr(),
                                                        ws.data_ptr(), st), fn)
                elif fn == 'wino4':
                    _lib.check(lib.ct_conv2d_wgrad_wino4(C.byref(d), dz.data_ptr(), cout, 0, dw4.data_ptr(),
                                                         ws.data_ptr(), st), fn)
                else:
                    _lib.check(lib.ct_conv2d_wgrad_wino4s(C.byref(d), dz.data_ptr(), cout, 0, dw4s.data_ptr(),
                                                          ws4s.data_ptr(), ws4s.numel(), st), fn)
            run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / a.iters)
        flops = 2.0 * a.batch * hw * hw * cin * cout * 9
        err = ((dw - dw2).abs().max() / dw.abs().max()).item()
        err4 = ((dw - dw4).abs().max() / dw.abs().max()).item()
        extra = ''
        if has4s:
            extra = '   F4 three-kernel %8.1f us (%5.1f TF alg) diff %.1e' % (res[3] * 1e3, flops / res[3] / 1e9,
                                                                             ((dw - dw4s).abs().max() / dw.abs().max()).item())
        print('%4d -> %4d @ %3d^2  direct %8.1f us (%5.1f TF)   F2 %8.1f us (%5.1f TF alg) diff %.1e   F4 %8.1f us (%5.1f TF alg) diff %.1e%s'
              % (cin, cout, hw, res[0] * 1e3, flops / res[0] / 1e9, res[1] * 1e3, flops / res[1] / 1e9, err,
                 res[2] * 1e3, flops / res[2] / 1e9, err4, extra), flush=True)


if __name__ == '__main__':
    main()
