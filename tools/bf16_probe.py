"""bf16 NHWC convolution (ct_conv2d_bf16_fwd): numerics against torch-CPU fp32 on bf16-rounded operands, and
timing on the RFBNet layer shapes (batch 32)."""
import argparse
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'context-transformer_amd'))
from ctdet import _lib  # noqa: E402

DEV = 'cuda:0'


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(lib, x, w, bias, stride, pad, dil, relu, iters=0):
    """x [B,Cin,H,W] fp32 (device), w [Cout,Cin,kh,kw] fp32 (device) -> y [B,Cout,OH,OW] fp32, avg ms."""
    B, Cin, H, W = x.shape
    Cout, _, kh, kw = w.shape
    OH = (H + 2 * pad - dil * (kh - 1) - 1) // stride + 1
    OW = (W + 2 * pad - dil * (kw - 1) - 1) // stride + 1
    cpad = (Cin + 7) // 8 * 8
    xb = torch.empty(B * H * W * cpad, dtype=torch.int16, device=DEV)
    _lib.check(lib.ct_nchw_f32_to_nhwc_bf16(x.data_ptr(), B, Cin, H * W, cpad, xb.data_ptr(), stream()), 'to nhwc')
    wp = torch.empty(lib.ct_conv_bf16_packed_elems(Cin, Cout, kh, kw), dtype=torch.int16, device=DEV)
    ptrs = (C.c_void_p * 1)(w.data_ptr())
    couts = (C.c_int * 1)(Cout)
    _lib.check(lib.ct_conv_pack_weights_bf16(ptrs, couts, 1, Cin, kh, kw, wp.data_ptr(), stream()), 'pack')
    scale = torch.ones(Cout, device=DEV)
    yb = torch.empty(B * OH * OW * Cout, dtype=torch.int16, device=DEV)
    d = _lib.ConvDesc()
    d.in_ = xb.data_ptr()
    d.batch, d.cin, d.h, d.w, d.in_ctot, d.in_coff = B, cpad, H, W, cpad, 0
    d.wpacked, d.scale, d.shift = wp.data_ptr(), scale.data_ptr(), bias.data_ptr()
    d.cout, d.kh, d.kw, d.stride, d.pad_h, d.pad_w, d.dil, d.oh, d.ow = Cout, kh, kw, stride, pad, pad, dil, OH, OW
    d.out, d.out_ctot, d.out_coff, d.relu = yb.data_ptr(), Cout, 0, int(relu)
    _lib.check(lib.ct_conv2d_bf16_fwd(C.byref(d), stream()), 'conv bf16')
    ms = 0.0
    if iters:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            lib.ct_conv2d_bf16_fwd(C.byref(d), stream())
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
    y = torch.empty(B, Cout, OH, OW, device=DEV)
    _lib.check(lib.ct_nhwc_bf16_to_nchw_f32(yb.data_ptr(), B, Cout, OH * OW, Cout, 0, y.data_ptr(), stream()), 'back')
    torch.cuda.synchronize()
    return y, ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    a = ap.parse_args()
    lib = _lib.lib()
    g = torch.Generator().manual_seed(0)
    # numerics: reference = fp32 conv of the bf16-rounded operands, output rounded to bf16
    for (B, Cin, H, W, Cout, k, s, p, dl) in [(2, 64, 19, 19, 96, 3, 1, 1, 1), (2, 40, 10, 11, 130, 3, 2, 1, 1),
                                              (1, 128, 19, 19, 64, 3, 1, 3, 3), (2, 256, 10, 10, 72, 1, 1, 0, 1),
                                              (2, 3, 30, 30, 64, 3, 1, 1, 1)]:
        x = torch.randn(B, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5
        b = torch.rand(Cout, generator=g) - 0.5
        want = F.relu(F.conv2d(x.bfloat16().float(), w.bfloat16().float(), b, s, p, dl)).bfloat16().float()
        got, _ = run(lib, x.to(DEV), w.to(DEV), b.to(DEV), s, p, dl, True)
        err = (got.cpu() - want).abs().max().item() / want.abs().max().item()
        print('numerics %s: max err / max |y| = %.2e (bf16 ulp = 3.9e-3)' % ((B, Cin, H, W, Cout, k, s, p, dl), err))
    for cin, cout, hw, k, dl in [(64, 64, 300, 3, 1), (128, 128, 150, 3, 1), (256, 256, 75, 3, 1), (512, 512, 38, 3, 1),
                                 (512, 512, 19, 3, 1), (512, 1024, 19, 3, 6), (1024, 1024, 19, 1, 1),
                                 (512, 960, 38, 1, 1), (256, 256, 10, 3, 1)]:
        x = torch.randn(a.batch, cin, hw, hw, device=DEV)
        w = torch.randn(cout, cin, k, k, device=DEV) * 0.05
        b = torch.zeros(cout, device=DEV)
        pad = dl * (k - 1) // 2
        _, ms = run(lib, x, w, b, 1, pad, dl, True, iters=10)
        flops = 2.0 * a.batch * hw * hw * cin * cout * k * k
        print('%4d -> %4d @ %3d^2 %dx%d d%d   %8.1f us   %7.1f TFLOP/s' % (cin, cout, hw, k, k, dl, ms * 1e3,
                                                                          flops / ms / 1e9))


if __name__ == '__main__':
    main()
