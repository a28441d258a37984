"""Winograd F(2x2,3x3) and F(4x4,3x3) convolutions (ct_conv2d_wino_fwd, ct_conv2d_wino4_fwd, F(2x2,3x3) on the
bf16 matrix pipe: ct_conv2d_wino_x3_fwd, eight-wave / two accumulators and four-wave / one accumulator, and F(4x4,3x3) as
transform / bf16x3 GEMM / transform kernels: ct_conv2d_wino4s_fwd, two / one accumulator) against torch-CPU conv2d
and against the direct implicit-GEMM kernel: same descriptor, same fused epilogues, 1e-4 relative (north_star's fp32
bar; F(4x4,3x3)'s own rounding is about 2e-5 of the output range at 512 input channels, checked below against fp64)."""
import zlib

import pytest
import torch

from conftest import rel_err
from ctdet import _lib, engine
from test_gpu_kernels import _bn, _ref_conv, _run_conv

pytestmark = pytest.mark.gpu
TOL = 1e-4
W = engine.WINO
VARIANTS = [engine.WINO, engine.WINO4, engine.WINOX, engine.WINO4S, engine.WINO4F, engine.WINO4H, engine.WINO4FH]
VIDS = ['f2x2', 'f4x4', 'f2x2_x3', 'f4x4_s', 'f4x4_f', 'f4x4_h', 'f4x4_fh']
X3V = (engine.WINOX, engine.WINO4S, engine.WINO4F, engine.WINO4H, engine.WINO4FH)         # cin must be a multiple of 16 (one MFMA k-group)
S3V = (engine.WINO4S, engine.WINO4H)                                     # the three-kernel forms (dilated layers too)


def _cin(W, cin):
    """Input channels of a test case for variant W: the bf16x3 kernel walks 16-channel chunks."""
    return (cin + 15) // 16 * 16 if W in X3V else cin


CASES = [  # name, B, Cin, H, W, Cout
    ('vgg', 2, 64, 38, 38, 128), ('odd_hw', 3, 16, 19, 17, 70), ('one_pixel', 2, 8, 1, 1, 5), ('tiny', 2, 24, 5, 5, 64),
    ('wide', 1, 128, 75, 75, 64), ('cout_156', 2, 32, 10, 10, 156), ('many_tiles', 4, 8, 150, 150, 8),
    ('deep', 1, 512, 19, 19, 96), ('row', 2, 16, 1, 9, 33), ('col', 2, 16, 7, 1, 33),
]


@pytest.mark.parametrize('W', VARIANTS, ids=VIDS)
@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_wino_matches_reference(case, W):
    name, B, Cin, H, Wd, Cout = case
    Cin = _cin(W, Cin)
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)
    x = torch.randn(B, Cin, H, Wd, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5
    b = torch.rand(Cout, generator=g) - 0.5
    want = _ref_conv(x, [(w, b, None, True)], 1, 1, 1)
    got = _run_conv(x, [(w, b, None, True)], 1, 1, 1, config=W)
    assert got.shape == want.shape and rel_err(got, want) < TOL
    direct = _run_conv(x, [(w, b, None, True)], 1, 1, 1, config=0)
    assert rel_err(got, direct) < TOL


DIL_CASES = [  # name, B, Cin, H, W, Cout, dilation -- conv6 (d 6 @19x19) and the RFB branches (d 2, 3, 5) of the network, odd shapes
    ('conv6', 2, 32, 19, 19, 48, 6), ('rfb_d2', 2, 16, 19, 19, 40, 2), ('rfb_d3', 1, 32, 38, 38, 33, 3), ('rfb_d5', 2, 16, 38, 37, 24, 5),
    ('d6_32', 1, 16, 32, 32, 16, 6), ('small_map', 3, 16, 5, 4, 8, 3), ('one_pixel', 2, 16, 1, 1, 5, 2), ('d8', 1, 16, 23, 9, 8, 8),
]


@pytest.mark.parametrize('W', list(S3V), ids=['f4x4_s', 'f4x4_h'])
@pytest.mark.parametrize('case', DIL_CASES, ids=[c[0] for c in DIL_CASES])
def test_wino4s_dilated_layers(case, W):
    """Dilated 3x3 layers (pad = dilation) on the three-kernel form: the d x d sub-lattices are tiled like small images
    (csrc/ct_wino4s.hip, wino4s_in_dil).  Against torch-CPU conv2d and the direct kernel, with BatchNorm + ReLU, into a channel
    slice of a wider buffer (how the RFB branches write their concat buffer)."""
    name, B, Cin, H, Wd, Cout, dil = case
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)
    x = torch.randn(B, Cin, H, Wd, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5
    parts = [(w, None, _bn(Cout, g), True)]
    want = _ref_conv(x, parts, 1, dil, dil)
    got = _run_conv(x, parts, 1, dil, dil, config=W, out_ctot=Cout + 11, out_coff=4)
    assert rel_err(got[:, 4:4 + Cout], want) < TOL
    assert torch.isnan(got[:, :4]).all() and torch.isnan(got[:, 4 + Cout:]).all()
    direct = _run_conv(x, parts, 1, dil, dil, config=0)
    assert rel_err(got[:, 4:4 + Cout], direct) < TOL
    # the fused kernels refuse the geometry
    with pytest.raises(_lib.CtdetError):
        _run_conv(x, parts, 1, dil, dil, config=engine.WINO4)


@pytest.mark.parametrize('W', VARIANTS, ids=VIDS)
def test_wino_fused_epilogues(W):
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 96, 19, 19, generator=g)
    # BN + ReLU, mixed ReLU parts in one launch (per-channel floor)
    w1 = torch.randn(40, 96, 3, 3, generator=g) * 0.05
    w2 = torch.randn(72, 96, 3, 3, generator=g) * 0.05
    parts = [(w1, None, _bn(40, g), True), (w2, None, _bn(72, g), False)]
    assert rel_err(_run_conv(x, parts, 1, 1, 1, config=W), _ref_conv(x, parts, 1, 1, 1)) < TOL
    # residual * scale + ReLU
    w3 = torch.randn(64, 96, 3, 3, generator=g) * 0.05
    bn3 = _bn(64, g)
    res = torch.randn(2, 64, 19, 19, generator=g)
    got = _run_conv(x, [(w3, None, bn3, True)], 1, 1, 1, config=W, res=res, res_scale=0.5)
    assert rel_err(got, _ref_conv(x, [(w3, None, bn3, True)], 1, 1, 1, res=res, res_scale=0.5)) < TOL
    # input channel slice [32:80), output at channel offset 8 of a 100-channel buffer, no ReLU
    w4 = torch.randn(64, 48, 3, 3, generator=g) * 0.05
    got = _run_conv(x, [(w4, None, bn3, False)], 1, 1, 1, config=W, cin_off=32, cin=48, out_ctot=100, out_coff=8)
    want = _ref_conv(x[:, 32:80], [(w4, None, bn3, False)], 1, 1, 1)
    assert rel_err(got[:, 8:72], want) < TOL
    assert torch.isnan(got[:, :8]).all() and torch.isnan(got[:, 72:]).all()


@pytest.mark.parametrize('W', VARIANTS, ids=VIDS)
def test_wino_rejects_other_geometries(W):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 16, 9, 9, generator=g)
    # the three-kernel form takes dilated layers with pad = dilation (test_wino4s_dilated_layers); pad != dilation stays out
    dilated = (3, 1, 1, 2, 16) if W in S3V else (3, 1, 2, 2, 16)
    for (k, stride, pad, dil, cin) in ((3, 2, 1, 1, 16), dilated, (1, 1, 0, 1, 16), (3, 1, 0, 1, 16)):
        w = torch.randn(8, cin, k, k, generator=g)
        with pytest.raises(_lib.CtdetError):
            _run_conv(x, [(w, None, None, False)], stride, pad, dil, config=W)
    with pytest.raises(_lib.CtdetError):                     # cin not a multiple of 8
        _run_conv(torch.randn(1, 3, 9, 9, generator=g), [(torch.randn(8, 3, 3, 3, generator=g), None, None, False)],
                  1, 1, 1, config=W)


@pytest.mark.parametrize('case', [(2, 16, 20, 20, 24, False, 1), (2, 16, 75, 75, 40, True, 0), (1, 8, 19, 17, 70, False, 0),
                                  (2, 8, 19, 17, 9, True, 1), (1, 8, 1, 1, 5, True, 0), (2, 8, 38, 38, 70, False, 1),
                                  (1, 16, 23, 18, 33, True, 1), (1, 16, 23, 18, 33, False, 0)])
@pytest.mark.parametrize('W', VARIANTS, ids=VIDS)
def test_wino_fused_maxpool(case, W):
    """MaxPool2d(2, 2[, ceil_mode]) behind the conv (models/RFB_Net_vgg.py:328-330) from the Winograd epilogue."""
    import torch.nn.functional as F
    B, Cin, H, Wd, Cout, ceil, full = case
    Cin = _cin(W, Cin)
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + Cout)
    x = torch.randn(B, Cin, H, Wd, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5
    b = torch.rand(Cout, generator=g) - 0.5
    be = engine.HipBackend('cuda:0')
    wp = torch.nn.Parameter(w.cuda(), requires_grad=False)
    bp = torch.nn.Parameter(b.cuda(), requires_grad=False)
    st = engine.ConvStep('t', [engine.ConvPart(wp, bp, None, True)], Cin, 3, 3, 1, 1, 1, 1, 'x', 0, H, Wd, 'y', 0)
    poh = -(-H // 2) if ceil else H // 2
    pow_ = -(-Wd // 2) if ceil else Wd // 2
    bufs = {'x': x.cuda(), 'y': torch.full((B, Cout, H, Wd), float('nan'), device='cuda')}
    pooled = torch.full((B, Cout, max(poh, 1), max(pow_, 1)), float('nan'), device='cuda')
    st.rt['config'] = W
    be.prepare_conv(st, bufs, B)
    if poh == 0 or pow_ == 0:
        return
    st.rt['pool'] = (pooled, poh, pow_, bool(full))
    be.run_conv(st)
    torch.cuda.synchronize()
    y = F.relu(F.conv2d(x, w, b, 1, 1))
    want = F.max_pool2d(y, 2, 2, 0, ceil_mode=ceil)
    assert rel_err(pooled.cpu(), want) < TOL
    if full:
        assert rel_err(bufs['y'].cpu(), y) < TOL
        assert torch.equal(F.max_pool2d(bufs['y'], 2, 2, 0, ceil_mode=ceil), pooled)     # same values, exactly
    else:
        assert torch.isnan(bufs['y']).all()                                             # full-resolution map skipped


@pytest.mark.parametrize('use_wino', VARIANTS + [0], ids=VIDS + ['direct'])
def test_head_scatter_output(use_wino):
    """Multibox head (models/RFB_Net_vgg.py:239-248): one fused loc|conf|obj conv writing channels-last into
    three flattened buffers at a prior offset -- Winograd and direct kernels against permute/view/cat."""
    g = torch.Generator().manual_seed(21)
    B, Cin, H, Wd, A, Cc = 2, 32, 10, 9, 6, 7
    x = torch.randn(B, Cin, H, Wd, generator=g)
    ws = [torch.randn(A * n, Cin, 3, 3, generator=g) * 0.06 for n in (4, Cc, 2)]
    bs = [torch.rand(A * n, generator=g) - 0.5 for n in (4, Cc, 2)]
    be = engine.HipBackend('cuda:0')
    parts = [engine.ConvPart(torch.nn.Parameter(w.cuda(), requires_grad=False),
                             torch.nn.Parameter(b.cuda(), requires_grad=False), None, False) for w, b in zip(ws, bs)]
    st = engine.ConvStep('h', parts, Cin, 3, 3, 1, 1, 1, 1, 'x', 0, H, Wd, None, 0)
    pbase, P = 11, 11 + H * Wd * A + 5                       # priors before / after this source
    st.segs = [engine.Segment('loc', 0, A * 4, A * 4, pbase * 4), engine.Segment('conf', A * 4, A * (4 + Cc), A * Cc, pbase * Cc),
               engine.Segment('obj', A * (4 + Cc), A * (6 + Cc), A * 2, pbase * 2)]
    bufs = {'x': x.cuda(), 'loc': torch.full((B, P * 4), float('nan'), device='cuda'),
            'conf': torch.full((B, P * Cc), float('nan'), device='cuda'), 'obj': torch.full((B, P * 2), float('nan'), device='cuda')}
    st.rt['config'] = use_wino
    be.prepare_conv(st, bufs, B)
    assert (st.rt.get('wino') or 0) == {**engine.WINO_TILE, 0: 0}[use_wino]
    be.run_conv(st)
    torch.cuda.synchronize()
    import torch.nn.functional as F
    for name, w, b, n in zip(('loc', 'conf', 'obj'), ws, bs, (4, Cc, 2)):
        want = F.conv2d(x, w, b, 1, 1).permute(0, 2, 3, 1).reshape(B, -1)
        got = bufs[name].cpu()
        lo, hi = pbase * n, pbase * n + H * Wd * A * n
        assert rel_err(got[:, lo:hi], want) < TOL, name
        assert torch.isnan(got[:, :lo]).all() and torch.isnan(got[:, hi:]).all(), name


@pytest.mark.parametrize('use_wino', VARIANTS + [0], ids=VIDS + ['direct'])
def test_conv_input_above_2gib_is_chunked(use_wino):
    """BASELINE configs[4] shapes put more than 2 GiB into one activation (32x64x512x512 fp32): the buffer
    descriptors are 32-bit, so both kernels split the batch -- the images on both sides of the split must be right."""
    import torch.nn.functional as F
    B, Cin, H, Cout = 33, 64, 512, 8
    g = torch.Generator().manual_seed(99)
    x = torch.randn(B, Cin, H, H, generator=g)
    assert x.numel() * 4 > 2 ** 31
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    b = torch.rand(Cout, generator=g) - 0.5
    got = _run_conv(x, [(w, b, None, True)], 1, 1, 1, config=use_wino)
    for n in (0, 30, 31, 32):                 # first image, both sides of the 2 GiB boundary, last image
        want = F.relu(F.conv2d(x[n:n + 1], w, b, 1, 1))
        assert rel_err(got[n:n + 1], want) < TOL, n


@pytest.mark.parametrize('W,bound', [(engine.WINO, 3e-6), (engine.WINO4, 5e-5), (engine.WINOX, 1e-6),
                                     (engine.WINO4S, 6e-6), (engine.WINO4F, 1.2e-5), (engine.WINO4H, 6e-6),
                                     (engine.WINO4FH, 1.2e-5)],
                         ids=VIDS)
def test_wino_rounding_error_vs_fp64(W, bound):
    """The transform-domain rounding of each variant on the deepest VGG shape (512 input channels, post-ReLU input):
    max error over the output range against an fp64 convolution.  Measured 1e-6 for F(2x2,3x3) and 2e-5 for
    F(4x4,3x3) (interpolation points 0, +-1, +-2, inf; transform entries up to 8).  The bf16x3 forms sum 16 channels
    inside an MFMA before one rounding; with two accumulators the large sum sees cin / 16 roundings.  The three-kernel
    F(4x4,3x3) form (bf16x3 GEMMs, output transform in double): measured 2.0e-6 (two accumulators)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 512, 38, 38, generator=g).relu()
    w = torch.randn(512, 512, 3, 3, generator=g) * (2.0 / (512 * 9)) ** 0.5
    b = torch.rand(512, generator=g) - 0.5
    want = F.relu(F.conv2d(x.double(), w.double(), b.double(), 1, 1))
    got = _run_conv(x, [(w, b, None, True)], 1, 1, 1, config=W)
    err = float((got.double() - want).abs().max() / want.abs().max())
    assert err < bound, err


def test_wino4_dgrad_weights():
    """ct_conv_pack_weights_wino4_dgrad: the data-gradient convolution (channels swapped, taps rotated) through the
    F(4x4,3x3) kernel equals autograd's input gradient of the forward convolution."""
    import ctypes as C
    import torch.nn.functional as F
    lib = _lib.lib()
    g = torch.Generator().manual_seed(8)
    B, Cin, Cout, H, Wd = 2, 24, 40, 13, 10
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1
    dy = torch.randn(B, Cout, H, Wd, generator=g)
    x = torch.zeros(B, Cin, H, Wd, requires_grad=True)
    F.conv2d(x, w, None, 1, 1).backward(dy)
    wd, dyd = w.cuda(), dy.cuda()
    U = torch.empty(lib.ct_conv_wino4_packed_floats(Cout, Cin), device='cuda')
    ptrs = (C.c_void_p * 1)(wd.data_ptr())
    couts = (C.c_int * 1)(Cout)
    _lib.check(lib.ct_conv_pack_weights_wino4_dgrad(ptrs, couts, 1, Cin, U.data_ptr(), None), 'pack')
    dx = torch.full((B, Cin, H, Wd), float('nan'), device='cuda')
    one, zero = torch.ones(Cin, device='cuda'), torch.zeros(Cin, device='cuda')
    d = _lib.ConvDesc()
    d.in_, d.batch, d.cin, d.h, d.w, d.in_ctot, d.in_coff = dyd.data_ptr(), B, Cout, H, Wd, Cout, 0
    d.cout, d.kh, d.kw, d.stride, d.pad_h, d.pad_w, d.dil = Cin, 3, 3, 1, 1, 1, 1
    d.oh, d.ow, d.out, d.out_ctot, d.out_coff = H, Wd, dx.data_ptr(), Cin, 0
    d.scale, d.shift = one.data_ptr(), zero.data_ptr()
    _lib.check(lib.ct_conv2d_wino4_fwd(C.byref(d), U.data_ptr(), None), 'wino4 dgrad')
    torch.cuda.synchronize()
    assert rel_err(dx.cpu(), x.grad) < TOL


def test_wino4_channel_split_matches_single_pass():
    """F(4x4,3x3) split over input channels (ct_conv_desc.ksplit): slabs + finishing kernel against the fused single
    pass and the reference, with BN, residual and ReLU in the finishing epilogue; two runs are bit-identical."""
    g = torch.Generator().manual_seed(31)
    x = torch.randn(2, 256, 19, 19, generator=g)
    w = torch.randn(72, 256, 3, 3, generator=g) * 0.03
    bn = _bn(72, g)
    res = torch.randn(2, 72, 19, 19, generator=g)
    want = _ref_conv(x, [(w, None, bn, True)], 1, 1, 1, res=res, res_scale=0.7)
    base = _run_conv(x, [(w, None, bn, True)], 1, 1, 1, config=engine.WINO4, res=res, res_scale=0.7, ksplit=0)
    assert rel_err(base, want) < TOL
    for ks in (2, 5, 8, -1):
        got = _run_conv(x, [(w, None, bn, True)], 1, 1, 1, config=engine.WINO4, res=res, res_scale=0.7, ksplit=ks)
        assert rel_err(got, want) < TOL, ks
        assert rel_err(got, base) < 1e-5, ks
        again = _run_conv(x, [(w, None, bn, True)], 1, 1, 1, config=engine.WINO4, res=res, res_scale=0.7, ksplit=ks)
        assert torch.equal(got, again), ks


@pytest.mark.parametrize('pool', [False, True], ids=['plain', 'fused_pool'])
def test_wino4_streamk_matches_plain_grid(pool, monkeypatch):
    """Stream-K form of the F(4x4,3x3) kernel (desc.ksplit = -2: persistent grid, last round cut by channel chunks,
    slabs + wino4_streamk_fixup) against the plain grid: 3.1 rounds of workgroups (800 items), BN + ReLU epilogue,
    optionally with the fused 2x2 max-pool -- the cut items' pooling happens in the fix-up kernel."""
    import ctypes as C
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(41)
    B, Cin, H, Wd, Cout = 8, 64, 76, 76, 256               # 19*19*8 = 2888 tiles -> 91 tile blocks x 4 cout blocks = 364
    x = torch.randn(B, Cin, H, Wd, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5
    b = torch.rand(Cout, generator=g) - 0.5
    y = F.relu(F.conv2d(x, w, b, 1, 1))
    want_pool = F.max_pool2d(y, 2, 2)
    monkeypatch.setenv('CTDET_W4_SK_FRAC', '1')            # cut the last round whatever its fill (1.42 rounds here)
    outs = []
    for ks in (0, -2, -2):
        be = engine.HipBackend('cuda:0')
        st = engine.ConvStep('t', [engine.ConvPart(torch.nn.Parameter(w.cuda(), requires_grad=False),
                                                   torch.nn.Parameter(b.cuda(), requires_grad=False), None, True)],
                             Cin, 3, 3, 1, 1, 1, 1, 'x', 0, H, Wd, 'y', 0)
        bufs = {'x': x.cuda(), 'y': torch.full((B, Cout, H, Wd), float('nan'), device='cuda')}
        st.rt['config'] = engine.WINO4
        be.prepare_conv(st, bufs, B)
        ws = torch.full((256 * 2 * 64 * 32 * 16,), float('nan'), device='cuda')
        d = st.rt['desc']
        d.ksplit, d.ksplit_ws, d.ksplit_ws_floats = ks, ws.data_ptr(), ws.numel()
        pooled = torch.full((B, Cout, H // 2, Wd // 2), float('nan'), device='cuda')
        if pool:
            st.rt['pool'] = (pooled, H // 2, Wd // 2, True)
        be.run_conv(st)
        torch.cuda.synchronize()
        assert rel_err(bufs['y'].cpu(), y) < TOL
        if pool:
            assert rel_err(pooled.cpu(), want_pool) < TOL
            assert torch.equal(F.max_pool2d(bufs['y'], 2, 2), pooled)
        outs.append(bufs['y'].clone())
    assert rel_err(outs[1].cpu(), outs[0].cpu()) < 1e-5         # other summation order in the cut items only
    assert not torch.equal(outs[1], outs[0])                     # ... so the stream-K path really ran
    assert torch.equal(outs[1], outs[2])                         # and is deterministic


def test_autotune_times_every_variant_and_keeps_a_correct_one(monkeypatch):
    """HipBackend.tune_conv (the path behind tools/tune_convs.py and CTDET_TUNE=1, which the test suite itself pins off):
    every direct tile, the bf16x3 tiles and the Winograd variants are timed on the layer's real buffers; whatever
    wins must still produce the reference result."""
    monkeypatch.setenv('CTDET_TUNE', '1')
    monkeypatch.setenv('CTDET_WINO_TILES', '2,4,23,44')        # default: 2,4,44,46 (+ the f16x2 twins on an f16x2 runtime)
    g = torch.Generator().manual_seed(77)
    B, Cin, H, Wd, Cout = 2, 64, 19, 19, 96
    x = torch.randn(B, Cin, H, Wd, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5
    b = torch.rand(Cout, generator=g) - 0.5
    be = engine.HipBackend('cuda:0')
    st = engine.ConvStep('t', [engine.ConvPart(torch.nn.Parameter(w.cuda(), requires_grad=False),
                                               torch.nn.Parameter(b.cuda(), requires_grad=False), None, True)],
                         Cin, 3, 3, 1, 1, 1, 1, 'x', 0, H, Wd, 'y', 0)
    bufs = {'x': x.cuda(), 'y': torch.full((B, Cout, H, Wd), float('nan'), device='cuda')}
    be.prepare_conv(st, bufs, B)
    best, times = be.tune_conv(st)
    finite = [t for t in times if t != float('inf')]
    assert len(finite) >= 4 + len(engine.wino_tiles(be, st)) and min(finite) > 0
    assert set(engine.wino_tiles(be, st)) == {2, 4, 23, 44}
    bufs['y'].fill_(float('nan'))
    be.run_conv(st)
    torch.cuda.synchronize()
    assert rel_err(bufs['y'].cpu(), _ref_conv(x, [(w, b, None, True)], 1, 1, 1)) < TOL


@pytest.mark.parametrize('W', [engine.WINO4H, engine.WINO4FH], ids=['f4x4_h', 'f4x4_fh'])
def test_f16x2_results_do_not_depend_on_batch_mates(W):
    """The f16x2 forms scale their operands by a power of two taken from a maximum of |input| (csrc/ct_f16x2.h); the subnormal lo
    pieces make the split depend on that exponent in the last bits, so the maximum is PER IMAGE: an image convolved alone and
    next to an image 4 096 times larger (another exponent for a batch-wide maximum) gives the same bits.  What the harness
    tests rely on (detections independent of batch composition and position)."""
    g = torch.Generator().manual_seed(77)
    x = torch.randn(3, 64, 38, 38, generator=g).relu()
    x[1] *= 4096.0
    x[2] *= 1.0 / 1024
    w = torch.randn(96, 64, 3, 3, generator=g) * 0.05
    b = torch.rand(96, generator=g) - 0.5
    together = _run_conv(x, [(w, b, None, True)], 1, 1, 1, config=W)
    for n in range(3):
        alone = _run_conv(x[n:n + 1], [(w, b, None, True)], 1, 1, 1, config=W)
        assert torch.equal(alone[0], together[n]), n
    want = _ref_conv(x, [(w, b, None, True)], 1, 1, 1)
    for n in range(3):                      # each image at ITS scale (a batch-wide scale would lose the small image's low bits)
        assert rel_err(together[n], want[n]) < TOL, n
