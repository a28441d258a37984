"""Winograd F(3x3, 2x2) and F(3x3, 4x4) weight gradients (ct_conv2d_wgrad_wino, ct_conv2d_wgrad_wino4, and the three-kernel
bf16x3 form ct_conv2d_wgrad_wino4s) against
autograd in float64 and against the direct weight-gradient kernel, through the C ABI.  What train.py:228 `losses.backward()` computes for the 3x3 / stride 1
weights of models/RFB_Net_vgg.py."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from ctdet import _lib

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
# (supported, workspace_bytes, run, tolerance vs float64): the large tile's transforms round 10x coarser
VARIANTS = {'f2': ('ct_conv_wgrad_wino_supported', 'ct_conv_wgrad_wino_workspace_bytes', 'ct_conv2d_wgrad_wino', 1e-5),
            'f4': ('ct_conv_wgrad_wino4_supported', 'ct_conv_wgrad_wino4_workspace_bytes', 'ct_conv2d_wgrad_wino4', 5e-5)}


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _desc(xd, B, Cin, H, W, ctot, coff, Cout, k=3, stride=1, pad=1, dil=1):
    d = _lib.ConvDesc()
    d.in_ = xd.data_ptr()
    d.batch, d.cin, d.h, d.w, d.in_ctot, d.in_coff = B, Cin, H, W, ctot, coff
    d.cout, d.kh, d.kw, d.stride, d.pad_h, d.pad_w, d.dil = Cout, k, k, stride, pad, pad, dil
    d.oh = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    d.ow = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    return d


GEOMS = [  # B, Cin, H, W, Cout, x slice (ctot, coff), dz slice (ctot, coff)
    (2, 64, 19, 19, 64, None, None),
    (3, 16, 10, 10, 40, None, None),            # partial channel blocks
    (2, 72, 19, 17, 130, None, None),           # odd sizes, ragged blocks both ways
    (1, 3, 30, 30, 16, None, None),             # first layer: 3 input channels
    (5, 8, 5, 5, 8, None, None),                # tiles straddle images inside a chunk
    (2, 24, 3, 3, 24, None, None), (4, 8, 1, 1, 8, None, None), (3, 8, 2, 2, 8, None, None),
    (2, 32, 38, 38, 64, (48, 9), (80, 7)),      # channel slices of wider buffers
    (2, 128, 75, 75, 64, None, None),           # many chunks per split, odd width
]


@pytest.mark.parametrize('variant', ['f2', 'f4'])
@pytest.mark.parametrize('g', GEOMS, ids=[str(i) for i in range(len(GEOMS))])
def test_wino_wgrad_vs_autograd(g, variant):
    B, Cin, H, W, Cout, xs, zs = g
    sup, wsb, run, tol = VARIANTS[variant]
    gen = torch.Generator().manual_seed(11 + Cin + H)
    xctot, xcoff = xs or (Cin, 0)
    zctot, zcoff = zs or (Cout, 0)
    xfull = torch.randn(B, xctot, H, W, generator=gen)
    dzfull = torch.randn(B, zctot, H, W, generator=gen)
    x = xfull[:, xcoff:xcoff + Cin].double().requires_grad_(True)
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x, w, None, 1, 1).backward(dzfull[:, zcoff:zcoff + Cout].double())
    lib = _lib.lib()
    xd, dzd = xfull.to(DEV), dzfull.to(DEV)
    d = _desc(xd, B, Cin, H, W, xctot, xcoff, Cout)
    assert getattr(lib, sup)(C.byref(d)) == 1
    ws = torch.empty(getattr(lib, wsb)(C.byref(d)) // 4, device=DEV)
    dw = torch.full((Cout, Cin, 3, 3), float('nan'), device=DEV)
    _lib.check(getattr(lib, run)(C.byref(d), dzd.data_ptr(), zctot, zcoff, dw.data_ptr(), ws.data_ptr(),
                                        _s()), 'wgrad wino')
    dw2 = torch.empty_like(dw)
    _lib.check(lib.ct_conv2d_wgrad(C.byref(d), dzd.data_ptr(), zctot, zcoff, dw2.data_ptr(), _s()), 'wgrad')
    torch.cuda.synchronize()
    e_w, e_d = rel_err(dw.cpu().double(), w.grad), rel_err(dw2.cpu().double(), w.grad)
    assert e_w < tol, (g, e_w, e_d)
    # a second call reuses the workspace (zeroed inside) and overwrites dw
    _lib.check(getattr(lib, run)(C.byref(d), dzd.data_ptr(), zctot, zcoff, dw2.data_ptr(), ws.data_ptr(),
                                        _s()), 'wgrad wino again')
    torch.cuda.synchronize()
    assert rel_err(dw2.cpu().double(), w.grad) < tol


GEOMS_4S = [  # B, Cin, H, W, Cout, x slice, dz slice -- cin % 16 == 0 (16-channel chunks of the bf16 fragments)
    (2, 64, 19, 19, 64, None, None), (3, 16, 10, 10, 40, None, None), (2, 80, 19, 17, 130, None, None),
    (5, 16, 5, 5, 8, None, None), (2, 32, 3, 3, 24, None, None), (4, 16, 1, 1, 8, None, None),
    (2, 32, 38, 38, 64, (48, 9), (80, 7)), (2, 128, 75, 75, 64, None, None), (2, 256, 38, 38, 156, None, None),
    # dilated (pad = dilation): conv6's d 6 at 19x19, the RFB branches' d 2 / 3 / 5, odd shapes -- 8th field
    (2, 32, 19, 19, 48, None, None, 6), (2, 16, 19, 19, 40, None, None, 2), (1, 32, 38, 37, 33, None, None, 3),
    (2, 16, 38, 38, 24, (20, 3), (30, 5), 5), (3, 16, 5, 4, 8, None, None, 3), (2, 16, 1, 1, 5, None, None, 2),
]


@pytest.mark.parametrize('g', GEOMS_4S, ids=[str(i) for i in range(len(GEOMS_4S))])
def test_wino4s_wgrad_vs_autograd(g):
    """ct_conv2d_wgrad_wino4s: F(3x3, 4x4) as transform kernels + the bf16x3 GEMM kernel of the forward form (k = tiles,
    split over workgroups, slabs added in order) against float64 autograd, held to the fused kernel's bound; two calls give
    bit-identical results (no atomics)."""
    B, Cin, H, W, Cout, xs, zs = g[:7]
    dil = g[7] if len(g) > 7 else 1
    gen = torch.Generator().manual_seed(11 + Cin + H)
    xctot, xcoff = xs or (Cin, 0)
    zctot, zcoff = zs or (Cout, 0)
    xfull = torch.randn(B, xctot, H, W, generator=gen)
    dzfull = torch.randn(B, zctot, H, W, generator=gen)
    x = xfull[:, xcoff:xcoff + Cin].double().requires_grad_(True)
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x, w, None, 1, dil, dil).backward(dzfull[:, zcoff:zcoff + Cout].double())
    lib = _lib.lib()
    xd, dzd = xfull.to(DEV), dzfull.to(DEV)
    d = _desc(xd, B, Cin, H, W, xctot, xcoff, Cout, pad=dil, dil=dil)
    assert lib.ct_conv_wgrad_wino4s_supported(C.byref(d)) == 1
    ws = torch.empty(lib.ct_conv_wgrad_wino4s_workspace_bytes(C.byref(d)), device=DEV, dtype=torch.uint8)
    outs = []
    for _ in range(2):
        ws.fill_(0xFF)                      # NaN bit patterns everywhere: nothing unwritten may reach the result
        dw = torch.full((Cout, Cin, 3, 3), float('nan'), device=DEV)
        _lib.check(lib.ct_conv2d_wgrad_wino4s(C.byref(d), dzd.data_ptr(), zctot, zcoff, dw.data_ptr(), ws.data_ptr(), ws.numel(),
                                              _s()), 'wgrad wino4s')
        torch.cuda.synchronize()
        outs.append(dw.cpu())
    assert rel_err(outs[0].double(), w.grad) < 5e-5, (g, rel_err(outs[0].double(), w.grad))
    assert torch.equal(outs[0], outs[1])
    # an undersized workspace is refused
    rc = lib.ct_conv2d_wgrad_wino4s(C.byref(d), dzd.data_ptr(), zctot, zcoff, dw.data_ptr(), ws.data_ptr(), 1024, _s())
    assert rc != 0


@pytest.mark.parametrize('variant', ['f2', 'f4'])
def test_wino_wgrad_rejects_other_geometries(variant):
    sup, wsb, run, tol = VARIANTS[variant]
    lib = _lib.lib()
    x = torch.zeros(1, 8, 10, 10, device=DEV)
    for kw in (dict(stride=2), dict(dil=2, pad=2), dict(k=1, pad=0), dict(pad=0)):
        d = _desc(x, 1, 8, 10, 10, 8, 0, 8, **kw)
        assert getattr(lib, sup)(C.byref(d)) == 0
        dw = torch.zeros(8 * 8 * 9, device=DEV)
        rc = getattr(lib, run)(C.byref(d), x.data_ptr(), 8, 0, dw.data_ptr(), dw.data_ptr(), _s())
        assert rc != 0


@pytest.mark.parametrize('variant', ['f2', 'f4'])
def test_wino_wgrad_above_2gib(variant):
    """An input buffer above 2 GiB (RFBNet-512 bs 32: conv1_1's output is exactly 2 GiB) goes through in batch
    chunks that accumulate into the same transform-domain workspace."""
    sup, wsb, run, tol = VARIANTS[variant]
    B, ctot, coff, Cin, Cout, S = 3, 704, 301, 8, 8, 512          # 3 x 704 x 512 x 512 x 4 B = 2.2 GB
    gen = torch.Generator(device=DEV).manual_seed(3)
    xfull = torch.randn(B, ctot, S, S, device=DEV, generator=gen)
    dz = torch.randn(B, Cout, S, S, device=DEV, generator=gen)
    x = xfull[:, coff:coff + Cin].cpu().double().requires_grad_(True)
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x, w, None, 1, 1).backward(dz.cpu().double())
    lib = _lib.lib()
    d = _desc(xfull, B, Cin, S, S, ctot, coff, Cout)
    assert getattr(lib, sup)(C.byref(d)) == 1
    ws = torch.empty(getattr(lib, wsb)(C.byref(d)) // 4, device=DEV)
    dw = torch.empty(Cout, Cin, 3, 3, device=DEV)
    _lib.check(getattr(lib, run)(C.byref(d), dz.data_ptr(), Cout, 0, dw.data_ptr(), ws.data_ptr(), _s()),
               'wgrad wino > 2 GiB')
    torch.cuda.synchronize()
    assert rel_err(dw.cpu().double(), w.grad) < tol
