"""Parity tests proper: every HIP kernel against the CPU oracle on the same seeded inputs.
All calls go through the C ABI (ctypes -> libctdet.so).  Run with `-m gpu` on an MI355X."""
import types
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from ctdet import _lib, engine, ops, synth
from oracle import box_ref, nms_ref, rfbnet_ref

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TOL = 1e-4      # north_star: fp32 conv/attention activations within 1e-4 (max|a-b| / max|b|)


@pytest.fixture(scope='module', autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail('the gpu tests need a HIP device; none visible')
    torch.set_num_threads(max(1, torch.get_num_threads()))
    info = ops.device_info(0)
    assert info['arch'].startswith('gfx950'), info


def _cuda(t):
    return t.to(DEV).contiguous()


# ------------------------------------------------------------------ convolution
class _Holder:
    """Minimal stand-ins so a single fused conv can be pushed through the real engine backend."""


def _run_conv(x, parts, stride, pad, dil, config=0, res=None, res_scale=1.0, cin_off=0, cin=None,
              out_ctot=None, out_coff=0, ksplit=None, x3=None):
    """parts: list of (weight, bias|None, bn_tuple|None, relu).  Returns the NCHW output tensor."""
    be = engine.HipBackend(DEV)
    cps = []
    for (w, b, bn, relu) in parts:
        wp = torch.nn.Parameter(_cuda(w), requires_grad=False)
        bp = torch.nn.Parameter(_cuda(b), requires_grad=False) if b is not None else None
        bnm = None
        if bn is not None:
            bnm = torch.nn.BatchNorm2d(w.shape[0], eps=1e-5).to(DEV)
            bnm.weight.data.copy_(bn[0]); bnm.bias.data.copy_(bn[1])
            bnm.running_mean.copy_(bn[2]); bnm.running_var.copy_(bn[3])
        cps.append(engine.ConvPart(wp, bp, bnm, relu))
    kh, kw = parts[0][0].shape[2:]
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    B, ctot, H, W = x.shape
    cin = cin if cin is not None else ctot - cin_off
    st = engine.ConvStep('t', cps, cin, kh, kw, stride, ph, pw, dil, 'x', cin_off, H, W, 'y', out_coff)
    cout = st.cout
    bufs = {'x': _cuda(x), 'y': torch.full((B, out_ctot or cout, st.oh, st.ow), float('nan'), device=DEV)}
    if res is not None:
        bufs['r'] = _cuda(res)
        st.res, st.res_coff, st.res_scale = 'r', 0, res_scale
    st.rt['config'] = config
    be.prepare_conv(st, bufs, B)
    if ksplit is not None:                      # explicit split-K factor (0 = off) instead of the library's choice
        st.rt['desc'].ksplit = ksplit
        st.rt['ksws'].fill_(float('nan'))       # the workspace needs no initialisation
    if x3 is not None:                          # bf16x3 tile config (ct_conv2d_x3_fwd)
        be.enable_x3(st, x3)
    be.run_conv(st)
    torch.cuda.synchronize()
    return bufs['y'].cpu()


def _ref_conv(x, parts, stride, pad, dil, res=None, res_scale=1.0):
    outs = []
    for (w, b, bn, relu) in parts:
        y = F.conv2d(x, w, b, stride, pad, dil)
        if bn is not None:
            y = F.batch_norm(y, bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5)
        if res is not None:
            y = y * res_scale + res
        outs.append(F.relu(y) if relu else y)
    return torch.cat(outs, 1)


def _bn(c, g):
    return (torch.rand(c, generator=g) * 0.4 + 0.8, torch.rand(c, generator=g) * 0.2 - 0.1,
            torch.rand(c, generator=g) * 0.2 - 0.1, torch.rand(c, generator=g) * 0.4 + 0.8)


CONV_CASES = [
    # name, B, Cin, H, W, Cout, k, stride, pad, dil
    ('vgg3x3', 2, 64, 38, 38, 128, 3, 1, 1, 1),
    ('first_cin3', 2, 3, 75, 75, 64, 3, 1, 1, 1),
    ('dil6', 2, 64, 19, 19, 160, 3, 1, 6, 6),
    ('dil3', 1, 128, 38, 38, 128, 3, 1, 3, 3),
    ('dil5_odd', 3, 34, 19, 17, 70, 3, 1, 5, 5),
    ('s2', 2, 128, 19, 19, 256, 3, 2, 1, 1),
    ('p0', 2, 128, 5, 5, 256, 3, 1, 0, 1),
    ('1x1', 2, 1024, 19, 19, 256, 1, 1, 0, 1),
    ('1x1s2', 2, 100, 19, 19, 96, 1, 2, 0, 1),
    ('1x3', 2, 64, 38, 38, 96, (1, 3), 1, (0, 1), 1),
    ('3x1', 2, 96, 38, 38, 128, (3, 1), 1, (1, 0), 1),
    ('4x4', 2, 128, 2, 2, 256, 4, 1, 1, 1),
    ('tiny1x1', 2, 256, 1, 1, 128, 1, 1, 0, 1),
]


@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_all_configs(case):
    name, B, Cin, H, W, Cout, k, stride, pad, dil = case
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)
    kh, kw = (k, k) if isinstance(k, int) else k
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, kh, kw, generator=g) * (2.0 / (Cin * kh * kw)) ** 0.5
    b = torch.rand(Cout, generator=g) - 0.5
    want = _ref_conv(x, [(w, b, None, True)], stride, pad, dil)
    ncfg = _lib.lib().ct_conv_num_configs()
    errs = {}
    names = [_lib.lib().ct_conv_config_name(i).decode() for i in range(ncfg)]
    for cfg in range(0, ncfg + 1):          # 0 = heuristic, 1.. = explicit tile configs
        if cfg and names[cfg - 1] == 'valu' and not (Cin == 3 and (kh, kw) == (3, 3) and Cout % 8 == 0):
            with pytest.raises(_lib.CtdetError):        # the vector-ALU kernel of the image layer refuses loudly
                _run_conv(x, [(w, b, None, True)], stride, pad, dil, config=cfg)
            continue
        got = _run_conv(x, [(w, b, None, True)], stride, pad, dil, config=cfg)
        errs[cfg] = rel_err(got, want)
    assert max(errs.values()) < TOL, errs
    assert name != 'first_cin3' or len(errs) == ncfg + 1


def test_conv_valu_image_layer():
    """conv_valu3x3_f32 (config 'valu', the 3-channel image layer on the vector ALU): strides, dilation, a ragged
    pixel count, BatchNorm scale + per-part ReLU floors, input / output channel slices -- against the torch reference
    and against the implicit-GEMM kernel on the same descriptor."""
    ncfg = _lib.lib().ct_conv_num_configs()
    valu = 1 + [_lib.lib().ct_conv_config_name(i).decode() for i in range(ncfg)].index('valu')
    g = torch.Generator().manual_seed(77)
    for (B, H, W, Cout, stride, pad, dil) in ((2, 75, 75, 64, 1, 1, 1), (3, 37, 41, 24, 2, 1, 1), (1, 33, 29, 40, 1, 2, 2),
                                              (1, 300, 300, 64, 1, 1, 1)):
        x = torch.randn(B, 3, H, W, generator=g) * 60
        w1 = torch.randn(Cout // 2, 3, 3, 3, generator=g) * 0.2
        w2 = torch.randn(Cout - Cout // 2, 3, 3, 3, generator=g) * 0.2
        parts = [(w1, None, _bn(Cout // 2, g), True), (w2, torch.rand(Cout - Cout // 2, generator=g), None, False)]
        want = _ref_conv(x, parts, stride, pad, dil)
        got = _run_conv(x, parts, stride, pad, dil, config=valu)
        assert rel_err(got, want) < TOL, (B, H, W, Cout, stride, pad, dil)
        assert rel_err(got, _run_conv(x, parts, stride, pad, dil, config=1)) < 2e-6
    x5 = torch.randn(2, 5, 40, 40, generator=g)
    w = torch.randn(16, 3, 3, 3, generator=g) * 0.2
    bn = _bn(16, g)
    got = _run_conv(x5, [(w, None, bn, True)], 1, 1, 1, cin_off=1, cin=3, out_ctot=30, out_coff=6, config=valu)
    assert torch.isnan(got[:, :6]).all() and torch.isnan(got[:, 22:]).all()       # untouched slices
    assert rel_err(got[:, 6:22], _ref_conv(x5[:, 1:4], [(w, None, bn, True)], 1, 1, 1)) < TOL


def test_conv_fused_epilogues():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 96, 19, 19, generator=g)
    # (a) BN + ReLU  (b) mixed ReLU parts in one launch  (c) residual*scale + ReLU  (d) channel slices
    w1 = torch.randn(40, 96, 1, 1, generator=g) * 0.1
    w2 = torch.randn(72, 96, 1, 1, generator=g) * 0.1
    bn1, bn2 = _bn(40, g), _bn(72, g)
    parts = [(w1, None, bn1, True), (w2, None, bn2, False)]
    got = _run_conv(x, parts, 1, 0, 1)
    assert rel_err(got, _ref_conv(x, parts, 1, 0, 1)) < TOL
    w3 = torch.randn(64, 96, 3, 3, generator=g) * 0.05
    bn3 = _bn(64, g)
    res = torch.randn(2, 64, 19, 19, generator=g)
    got = _run_conv(x, [(w3, None, bn3, True)], 1, 2, 2, res=res, res_scale=0.5)
    assert rel_err(got, _ref_conv(x, [(w3, None, bn3, True)], 1, 2, 2, res=res, res_scale=0.5)) < TOL
    # input channel slice [32:80) and output written at channel offset 8 of a 100-channel buffer
    w4 = torch.randn(64, 48, 3, 3, generator=g) * 0.05
    got = _run_conv(x, [(w4, None, bn3, False)], 1, 1, 1, cin_off=32, cin=48, out_ctot=100, out_coff=8)
    want = _ref_conv(x[:, 32:80], [(w4, None, bn3, False)], 1, 1, 1)
    assert rel_err(got[:, 8:72], want) < TOL
    assert torch.isnan(got[:, :8]).all() and torch.isnan(got[:, 72:]).all()     # untouched slices


def test_conv_split_k():
    """Split-K (ct_conv_desc.ksplit) against the single-pass kernel and the reference, all epilogue kinds."""
    g = torch.Generator().manual_seed(21)
    x = torch.randn(2, 200, 5, 5, generator=g)
    w = torch.randn(72, 200, 3, 3, generator=g) * 0.03
    bn = _bn(72, g)
    res = torch.randn(2, 72, 5, 5, generator=g)
    want = _ref_conv(x, [(w, None, bn, True)], 1, 1, 1, res=res, res_scale=0.7)
    base = _run_conv(x, [(w, None, bn, True)], 1, 1, 1, res=res, res_scale=0.7, ksplit=0)
    assert rel_err(base, want) < TOL
    for ks in (2, 3, 7, 1000, -1):              # 1000 > number of k-steps: clamped; -1: library's choice
        for cfg in (0, 4, 5):
            got = _run_conv(x, [(w, None, bn, True)], 1, 1, 1, res=res, res_scale=0.7, ksplit=ks, config=cfg)
            assert rel_err(got, want) < TOL, (ks, cfg)
            again = _run_conv(x, [(w, None, bn, True)], 1, 1, 1, res=res, res_scale=0.7, ksplit=ks, config=cfg)
            assert torch.equal(got, again), 'split-K must be run-to-run deterministic'  
    # 1x1 stride 2, mixed ReLU parts, output slice of a wider buffer
    x2 = torch.randn(3, 512, 10, 10, generator=g)
    w1 = torch.randn(40, 512, 1, 1, generator=g) * 0.05
    w2 = torch.randn(24, 512, 1, 1, generator=g) * 0.05
    parts = [(w1, None, _bn(40, g), True), (w2, None, _bn(24, g), False)]
    got = _run_conv(x2, parts, 2, 0, 1, out_ctot=80, out_coff=5, ksplit=4)
    assert rel_err(got[:, 5:69], _ref_conv(x2, parts, 2, 0, 1)) < TOL
    assert torch.isnan(got[:, :5]).all() and torch.isnan(got[:, 69:]).all()


def test_maxpool_variants():
    g = torch.Generator().manual_seed(3)
    # (19, 19) / (32, 32) / odd shapes with k 3, stride 1, pad 1: the plane kernel of pool5 (one plane, odd plane counts,
    # more than 256 and more than 2 x 256 elements per plane); 65 x 65 > 4096 elements: the generic kernel
    for (H, W, k, s, p, ceil) in [(300, 300, 2, 2, 0, False), (75, 75, 2, 2, 0, True), (19, 19, 3, 1, 1, False),
                                  (38, 37, 2, 2, 0, True), (32, 32, 3, 1, 1, False), (1, 1, 3, 1, 1, False),
                                  (7, 23, 3, 1, 1, False), (64, 64, 3, 1, 1, False), (65, 65, 3, 1, 1, False)]:
        x = torch.randn(2, 5, H, W, generator=g)
        got = ops.maxpool2d(_cuda(x), k, s, p, ceil).cpu()
        want = F.max_pool2d(x, k, s, p, ceil_mode=ceil)
        assert got.shape == want.shape and torch.equal(got, want), (H, W, k, s, p, ceil)
    x = torch.randn(1, 1, 19, 19, generator=g)                 # a single plane
    assert torch.equal(ops.maxpool2d(_cuda(x), 3, 1, 1, False).cpu(), F.max_pool2d(x, 3, 1, 1))
    x = torch.randn(3, 683, 19, 19, generator=g)               # 2 049 planes: ragged last workgroup
    assert torch.equal(ops.maxpool2d(_cuda(x), 3, 1, 1, False).cpu(), F.max_pool2d(x, 3, 1, 1))


# ------------------------------------------------------------------ boxes
def test_decode_encode_detect_softmax():
    priors = box_ref.prior_box(box_ref.ANCHOR_CFGS['VOC_300'])
    P = priors.shape[0]
    g = torch.Generator().manual_seed(7)
    loc = torch.randn(2, P, 4, generator=g)
    got = ops.decode(_cuda(loc), _cuda(priors), [0.1, 0.2]).cpu()
    want = torch.stack([box_ref.decode(loc[i], priors, [0.1, 0.2]) for i in range(2)])
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=2e-6, atol=1e-7)   # expf 1-ulp class
    scale = torch.tensor([500., 375., 500., 375.])
    got = ops.decode(_cuda(loc), _cuda(priors), [0.1, 0.2], _cuda(scale)).cpu()
    np.testing.assert_allclose(got.numpy(), (want * scale).numpy(), rtol=2e-6, atol=1e-4)   # pixels: |x| <= ~1e3
    matched = point = box_ref.point_form(priors) + 0.01
    got = ops.encode(_cuda(matched), _cuda(priors), [0.1, 0.2]).cpu()
    want = box_ref.encode(matched, priors, [0.1, 0.2])
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-5, atol=2e-5)
    conf_l = torch.randn(2, P, 20, generator=g) * 2
    obj_l = torch.randn(2, P, 2, generator=g)
    conf, obj = torch.softmax(conf_l, -1), torch.softmax(obj_l, -1)
    np.testing.assert_allclose(ops.softmax_lastdim(_cuda(conf_l)).cpu().numpy(), conf.numpy(), rtol=1e-5, atol=1e-7)
    wb, ws = box_ref.detect(loc, conf, obj, priors)
    gb, gs = ops.detect_fused(_cuda(loc), _cuda(conf), _cuda(obj), _cuda(priors), [0.1, 0.2], False)
    np.testing.assert_allclose(gb.cpu().numpy(), wb.numpy(), rtol=2e-6, atol=1e-7)
    assert torch.equal(gs.cpu(), ws)                       # products of identical fp32 inputs: bit-exact
    gb, gs = ops.detect_fused(_cuda(loc), _cuda(conf_l), _cuda(obj_l), _cuda(priors), [0.1, 0.2], True)
    np.testing.assert_allclose(gs.cpu().numpy(), ws.numpy(), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize('C', [15, 20, 60, 64, 65])
@pytest.mark.parametrize('softmax', [False, True])
def test_detect_kernels_agree_bit_for_bit(C, softmax):
    """ct_detect_fused has two kernels (csrc/ct_box.hip): the row-in-registers one (up to 64 classes, 16-byte aligned conf and
    scores: linear 16-byte copies, no integer divisions) and the strided-copy one (any class count, any alignment).  Same
    expressions in the same order: identical bits -- checked by handing the same values over once aligned and once through a view
    that starts 4 bytes into its buffer (ragged last workgroup: 2 x 11 620 rows)."""
    priors = box_ref.prior_box(box_ref.ANCHOR_CFGS['VOC_300'])
    P = priors.shape[0]
    g = torch.Generator().manual_seed(100 + C)
    loc = _cuda(torch.randn(2, P, 4, generator=g))
    conf = torch.randn(2, P, C, generator=g) * 3
    obj = torch.randn(2, P, 2, generator=g)
    if not softmax:
        conf, obj = torch.softmax(conf, -1), torch.softmax(obj, -1)
    conf, obj = _cuda(conf), _cuda(obj)
    gb, gs = ops.detect_fused(loc, conf, obj, _cuda(priors), [0.1, 0.2], softmax)
    raw = torch.empty(conf.numel() + 1, device=conf.device)
    shifted = raw[1:].view(2, P, C)
    shifted.copy_(conf)
    assert shifted.data_ptr() % 16 == 4 and conf.data_ptr() % 16 == 0
    hb, hs = ops.detect_fused(loc, shifted, obj, _cuda(priors), [0.1, 0.2], softmax)
    assert torch.equal(gb, hb) and torch.equal(gs, hs)
    wb, ws = box_ref.detect(loc.cpu(), conf.cpu() if not softmax else torch.softmax(conf.cpu(), -1),
                            obj.cpu() if not softmax else torch.softmax(obj.cpu(), -1), priors)
    np.testing.assert_allclose(gs.cpu().numpy(), ws.numpy(), rtol=1e-5, atol=1e-7)


def test_jaccard_and_match_exact(golden):
    g = golden('box_ops.npz')
    priors = box_ref.prior_box(box_ref.ANCHOR_CFGS['VOC_300'])
    truths = torch.from_numpy(g['match_truths'])
    labels = torch.from_numpy(g['match_labels'])
    ov = ops.jaccard(_cuda(truths), _cuda(box_ref.point_form(priors))).cpu()
    want = box_ref.jaccard(truths, box_ref.point_form(priors))
    assert torch.equal(ov, want)                            # same fp32 expression, no contraction
    ov2 = ops.jaccard(_cuda(truths), _cuda(priors), b_center_form=True).cpu()
    assert torch.equal(ov2, want)
    tg = torch.cat([truths, labels], 1)
    rng = np.random.RandomState(0)
    batch = [tg, tg[:2], tg[[3, 4]], synth.targets(1, 21, 5)[0]]
    for thr in (0.5, 0.35):
        loc_t, conf_t, obj_t, ovl = ops.match_batched([_cuda(t) for t in batch], _cuda(priors), thr, [0.1, 0.2], True)
        for i, t in enumerate(batch):
            wl, wc, wo, wov = box_ref.match(thr, t[:, :4], priors, [0.1, 0.2], t[:, 4:])
            assert torch.equal(conf_t[i].cpu(), wc), (thr, i)
            assert torch.equal(obj_t[i].cpu(), wo), (thr, i)
            assert torch.equal(ovl[i].cpu(), wov), (thr, i)
            np.testing.assert_allclose(loc_t[i].cpu().numpy(), wl.numpy(), rtol=1e-5, atol=2e-5)
    # the in-place reference signature
    from utils import box_utils as bu
    P = priors.shape[0]
    loc_t = torch.zeros(1, P, 4, device=DEV); conf_t = torch.zeros(1, P, 2, device=DEV)
    obj_t = torch.zeros(1, P, dtype=torch.bool, device=DEV)
    bu.match(0.5, _cuda(truths), _cuda(priors), [0.1, 0.2], _cuda(labels), loc_t, conf_t, obj_t, 0)
    assert np.array_equal(conf_t[0].cpu().numpy(), g['match50_conf'])
    assert np.array_equal(obj_t[0].cpu().numpy(), g['match50_obj'])


# ------------------------------------------------------------------ NMS
def test_nms_goldens_bit_exact(golden):
    g = golden('nms.npz')
    from utils.nms_wrapper import nms
    for ci in range(int(g['ncases'])):
        d = g['c%d_dets' % ci]
        for thr in (0.45, 0.3, 0.5, 0.7):
            tag = 'c%d_t%02d' % (ci, int(round(thr * 100)))
            assert [int(i) for i in nms(d, thr)] == list(g[tag + '_gt']), tag
            order = nms_ref.stable_desc_order(d[:, 4])
            kge = ops.nms_sorted_host(d[order], thr, ge=True)
            assert list(order[kge]) == list(nms_ref.nms(d, thr, ge=True)), tag
    assert [int(i) for i in nms(g['ka_dets'], 0.45)] == [0, 2]
    assert [int(i) for i in nms(g['eq_dets'], 0.5)] == [0, 1]            # IoU == thresh: '>' keeps
    assert [int(i) for i in nms(g['eq_dets'], 0.5, force_cpu=True)] == [0]
    order = nms_ref.stable_desc_order(g['eq_dets'][:, 4])
    assert list(ops.nms_sorted_host(g['eq_dets'][order], 0.5, ge=True)) == [0]


@pytest.mark.parametrize('n', [1, 63, 64, 65, 255, 256, 257, 1000, 5000, 11620])
def test_nms_sizes_vs_c_oracle(n):
    rng = np.random.RandomState(n)
    d = synth.clustered_dets(n, clusters=max(2, n // 40), rng=rng)
    order = nms_ref.stable_desc_order(d[:, 4])
    ds = d[order]
    for thr, ge in ((0.45, False), (0.45, True), (0.1, False), (0.9, False)):
        got = ops.nms_sorted_host(ds, thr, ge=ge)
        want = nms_ref.nms_sorted_c(ds, thr, ge=ge)
        assert np.array_equal(got, want), (n, thr, ge, len(got), len(want))


def test_nms_all_disjoint_exceeds_lds_list():
    # 6000 disjoint boxes: every box is kept -> kept list spills past the 2048-entry LDS window
    n = 6000
    i = np.arange(n)
    xy = np.stack([(i % 100) * 20.0, (i // 100) * 20.0], 1)
    d = np.concatenate([xy, xy + 9.0, np.linspace(0.99, 0.02, n)[:, None]], 1).astype(np.float32)
    got = ops.nms_sorted_host(d, 0.45)
    assert np.array_equal(got, np.arange(n))
    # heavy overlap: everything suppressed by the first
    d2 = np.tile(np.array([[10, 10, 60, 60, 0.5]], np.float32), (3000, 1))
    d2[:, 4] = np.linspace(0.9, 0.1, 3000)
    assert list(ops.nms_sorted_host(d2, 0.45)) == [0]


def test_nms_batched_segments():
    rng = np.random.RandomState(9)
    lens = [0, 1, 300, 64, 0, 777, 2, 1500]
    segs = []
    for n in lens:
        d = synth.clustered_dets(n, rng=rng) if n else np.zeros((0, 5), np.float32)
        segs.append(d[nms_ref.stable_desc_order(d[:, 4])] if n else d)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    dets = torch.from_numpy(np.concatenate(segs, 0))
    keep, cnt = ops.nms_batched(_cuda(dets), _cuda(torch.from_numpy(off)), 0.45)
    keep, cnt = keep.cpu().numpy(), cnt.cpu().numpy()
    for s, n in enumerate(lens):
        want = nms_ref.nms_sorted_c(segs[s], 0.45)
        assert cnt[s] == len(want), s
        assert np.array_equal(keep[off[s]:off[s] + cnt[s]], want), s


def test_box_utils_nms_plain_iou(golden):
    from utils import box_utils as bu
    g = golden('box_ops.npz')
    priors = box_ref.prior_box(box_ref.ANCHOR_CFGS['VOC_300'])
    P = priors.shape[0]
    gen = torch.Generator().manual_seed(11)
    loc = torch.randn(2, P, 4, generator=gen)
    conf = torch.softmax(torch.randn(2, P, 20, generator=gen) * 2, -1)
    obj = torch.softmax(torch.randn(2, P, 2, generator=gen), -1)
    boxes, _ = box_ref.detect(loc, conf, obj, priors)       # oracle boxes (bit-exact to the reference)
    sc = torch.rand(P, generator=gen)
    for (ovt, topk) in ((0.5, 200), (0.3, 400)):
        keep, count = bu.nms(_cuda(boxes[0]), _cuda(sc), ovt, topk)
        assert np.array_equal(keep[:count].cpu().numpy(), g['bunms_%02d_%d_keep' % (int(ovt * 100), topk)])


# ------------------------------------------------------------------ batched post-processing
def _post_inputs(seed, B, T, P, priors):
    gen = torch.Generator().manual_seed(seed)
    loc = torch.randn(B, P, 4, generator=gen) * 0.5
    conf = torch.softmax(torch.randn(B, P, T, generator=gen) * 3.0, -1)
    obj = torch.softmax(torch.randn(B, P, 2, generator=gen) * 2.0 + torch.tensor([2.5, 0.0]), -1)
    return loc, conf, obj


def test_postprocess_matches_reference_pipeline(golden):
    g = golden('pipeline.npz')
    priors = box_ref.prior_box(box_ref.ANCHOR_CFGS['VOC_300'])
    P, B, T = priors.shape[0], 2, 20
    loc, conf, obj = _post_inputs(int(g['seed']), B, T, P, priors)
    boxes, scores = box_ref.detect(loc, conf, obj, priors)          # reference-exact inputs
    scale = torch.tensor([500., 375., 500., 375.])
    pp = ops.PostProcessor(B, P, T, DEV)
    pp.run(_cuda(boxes * scale), _cuda(scores))
    allb = pp.to_all_boxes()
    for i in range(B):
        # bit-exact against the oracle pipeline evaluated HERE on the same inputs ...
        want = nms_ref.postprocess_image(boxes[i].numpy(), scores[i].numpy(), (500, 375), nms_fn=nms_ref.nms_c)
        for j in range(1, T + 1):
            assert np.array_equal(allb[i][j], want[j]), (i, j)
            # ... and equal to the rows captured from the reference up to the last-ulp differences
            # torch.exp shows between host CPUs (the golden was made on another machine)
            ref = g['img%d_cls%d' % (i, j)]
            assert allb[i][j].shape == ref.shape, (i, j)
            np.testing.assert_allclose(allb[i][j], ref, rtol=1e-5, atol=1e-4)


def test_postprocess_all_pass_regime_vs_oracle():
    """R1: every prior passes the 0.01 threshold -> 11620 candidates per class."""
    priors = box_ref.prior_box(box_ref.ANCHOR_CFGS['VOC_300'])
    P, B, T = priors.shape[0], 2, 3
    gen = torch.Generator().manual_seed(77)
    loc = torch.randn(B, P, 4, generator=gen) * 0.3
    conf = torch.softmax(torch.randn(B, P, T, generator=gen) * 0.3, -1)
    obj = torch.softmax(torch.randn(B, P, 2, generator=gen) * 0.3, -1)
    boxes, scores = box_ref.detect(loc, conf, obj, priors)
    scale = torch.tensor([500., 375., 500., 375.])
    bs = (boxes * scale)
    pp = ops.PostProcessor(B, P, T, DEV, out_cap=P)
    pp.run(_cuda(bs), _cuda(scores), max_per_image=200)
    allb = pp.to_all_boxes()
    for i in range(B):
        want = nms_ref.postprocess_image(boxes[i].numpy(), scores[i].numpy(), (500, 375), nms_fn=nms_ref.nms_c)
        assert int((scores[i, :, 1:] > 0.01).sum()) > 0.9 * P * T
        for j in range(1, T + 1):
            assert np.array_equal(allb[i][j], want[j]), (i, j, allb[i][j].shape, want[j].shape)
    pp.run(_cuda(bs), _cuda(scores), max_per_image=0)                # no top-k rule
    allb = pp.to_all_boxes()
    for i in range(B):
        want = nms_ref.postprocess_image(boxes[i].numpy(), scores[i].numpy(), (500, 375), max_per_image=0,
                                         nms_fn=nms_ref.nms_c)
        for j in range(1, T + 1):
            assert np.array_equal(allb[i][j], want[j]), (i, j)


@pytest.mark.parametrize('regime', ['identical_boxes', 'two_levels', 'one_bin', 'few'])
def test_postprocess_partial_sort_paths_vs_oracle(regime):
    """ct_postprocess_batched sorts only the best >= 1024 candidates of a class when the top-k rule is on (csrc/ct_post.hip,
    select_sort_kernel<.., true>) and re-sorts in full the segments whose cut falls behind that prefix.  The regimes force
    each path: identical boxes -> one box kept per class, fewer than max_per_image kept in the bounding pass -> every
    segment flagged and sorted in full; two score levels -> the cut bin alone overflows the LDS buffer even after the
    second histogram -> full sort inside the first launch; all scores inside one histogram bin -> the 11-bit refinement;
    few -> fewer candidates than the prefix (the trained-detector case).  Bit-exact against the oracle loop
    (test.py:136-161), ties in ascending prior order."""
    priors = box_ref.prior_box(box_ref.ANCHOR_CFGS['VOC_300'])
    P, B, T = priors.shape[0], 2, 3
    gen = torch.Generator().manual_seed(91)
    pb = torch.cat([priors[:, :2] - priors[:, 2:] / 2, priors[:, :2] + priors[:, 2:] / 2], 1).clamp(0, 1)
    boxes = pb[None].repeat(B, 1, 1).contiguous()
    scores = torch.zeros(B, P, T + 1)
    if regime == 'identical_boxes':
        boxes[:] = torch.tensor([0.2, 0.3, 0.6, 0.7])
        scores[:, :, 1:] = 0.02 + 0.9 * torch.rand(B, P, T, generator=gen)
    elif regime == 'two_levels':
        scores[:, :, 1:] = torch.where(torch.rand(B, P, T, generator=gen) < 0.5, torch.tensor(0.25), torch.tensor(0.75))
    elif regime == 'one_bin':
        scores[:, :, 1:] = 0.5 + 0.06 * torch.rand(B, P, T, generator=gen)
    else:
        for b in range(B):
            for c in range(T):
                idx = torch.randperm(P, generator=gen)[:300 + 100 * c]
                scores[b, idx, 1 + c] = 0.02 + 0.9 * torch.rand(idx.numel(), generator=gen)
    scale = torch.tensor([500., 375., 500., 375.])
    pp = ops.PostProcessor(B, P, T, DEV, out_cap=P)
    for topk in (200, 0):
        pp.run(_cuda(boxes * scale), _cuda(scores), max_per_image=topk)
        allb = pp.to_all_boxes()
        for i in range(B):
            want = nms_ref.postprocess_image(boxes[i].numpy(), scores[i].numpy(), (500, 375), max_per_image=topk,
                                             nms_fn=nms_ref.nms_c)
            for j in range(1, T + 1):
                assert np.array_equal(allb[i][j], want[j]), (regime, topk, i, j, allb[i][j].shape, want[j].shape)


# ------------------------------------------------------------------ context attention
@pytest.mark.parametrize('form', ['bf16x3', 'f16x2'])
@pytest.mark.parametrize('setting,d,T', [('transfer', 60, 20), ('incre', 15, 5)])
def test_ctx_attention_vs_oracle(setting, d, T, form, monkeypatch):
    """Both operand forms of the inference forward: bf16x3 (default) and the opt-in f16x2 one (CTDET_ATTN_H2=1: a query row
    scaled by its own power of two, phi / g per image, probabilities by 2^14)."""
    monkeypatch.setenv('CTDET_ATTN_H2', '1' if form == 'f16x2' else '0')
    assert ops.lib().ct_ctx_attention_piece_products() == (3 if form == 'f16x2' else 6)
    B, P, M = 2, 11620, 1858
    shapes = {k: v for k, v in rfbnet_ref.param_shapes(300, d, 2, 'ours', setting).items()
              if k.split('.')[0] in ('theta', 'phi', 'g', 'Wz', 'OBJ_Target', 'scale', 'fc_base')}
    sd = synth.fill_state_dict(shapes)
    if setting == 'incre':
        sd['fc_base.weight'] = torch.randn(d, d) * 0.2
    gen = torch.Generator().manual_seed(3)
    conf = torch.randn(B, P, d, generator=gen) * 1.5
    pool = torch.randn(B, M, d, generator=gen) * 1.5
    pool[0, 17] *= 6.0              # a spiking key: forces large online-softmax rescales
    want = rfbnet_ref.context_block(sd, conf, pool, setting)
    prm = dict(theta_w=sd['theta.weight'], theta_b=sd['theta.bias'], phi_w=sd['phi.weight'], phi_b=sd['phi.bias'],
               g_w=sd['g.weight'], g_b=sd['g.bias'], wz=sd['Wz'], obj_w=sd['OBJ_Target.weight'])
    if setting == 'incre':
        prm.update(fc_w=sd['fc_base.weight'], fc_b=sd['fc_base.bias'])
    prm = {k: _cuda(v) for k, v in prm.items()}
    prm['scale'] = 5.0
    got = ops.ctx_attention(_cuda(conf), _cuda(pool), prm, setting == 'incre').cpu()
    assert got.shape == want.shape
    assert rel_err(got, want) < TOL, rel_err(got, want)


def test_degenerate_inputs():
    """Edge cases of the reference's own loops: an image whose scores never pass the threshold
    (test.py:141-143 writes empty [0,5] arrays), images without ground truth in `match`, a single box in NMS."""
    priors = box_ref.prior_box(box_ref.ANCHOR_CFGS['VOC_300'])
    P = priors.shape[0]
    B, T = 2, 20
    boxes = torch.rand(B, P, 4)
    scores = torch.full((B, P, T + 1), 0.001)               # everything below conf_thresh = 0.01
    scores[1, 5, 3] = 0.9                                     # ... except one prior of image 1, class 3
    post = ops.PostProcessor(B, P, T, DEV)
    post.run(_cuda(boxes), _cuda(scores), 0.01, 0.45, False, 200)
    res = post.to_all_boxes()
    assert all(res[0][j].shape == (0, 5) and res[0][j].dtype == np.float32 for j in range(T + 1))
    assert res[1][3].shape == (1, 5) and abs(float(res[1][3][0, 4]) - 0.9) < 1e-7
    assert np.array_equal(res[1][3][0, :4], boxes[1, 5].numpy())
    assert sum(len(res[1][j]) for j in range(T + 1)) == 1
    # match: an image without objects gives all-background targets (labels 0, weights 1, loc finite)
    t0 = torch.zeros(0, 6)
    t1 = torch.tensor([[0.1, 0.1, 0.5, 0.6, 3.0, 1.0]])
    loc_t, conf_t, obj_t = ops.match_batched([_cuda(t0), _cuda(t1)], _cuda(priors), 0.5, (0.1, 0.2))
    assert not obj_t[0].any() and (conf_t[0, :, 0] == 0).all() and (conf_t[0, :, 1] == 1).all()
    assert obj_t[1].any() and torch.isfinite(loc_t[1]).all()
    want = box_ref.match(0.5, t1[:, :4], priors, [0.1, 0.2], t1[:, 4:])
    assert torch.equal(conf_t[1].cpu(), want[1]) and torch.equal(obj_t[1].cpu(), want[2])
    # one box: kept
    one = np.array([[10, 10, 50, 60, 0.7]], np.float32)
    assert ops.nms_sorted_host(one, 0.45).tolist() == [0]


def test_c_caller_runs_the_nms_contract_on_the_device(tmp_path):
    """examples/c_caller.c (plain C, no Python) through ct_nms_sorted_host: the drop-in for the reference's `_nms`."""
    import os
    import shutil
    import subprocess
    if shutil.which('gcc') is None:
        pytest.skip('gcc not available')
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(repo, 'context-transformer_amd', 'lib')
    exe = str(tmp_path / 'c_caller')
    subprocess.check_call(['gcc', '-std=c99', '-I' + os.path.join(repo, 'include'), os.path.join(repo, 'examples', 'c_caller.c'),
                           '-L' + libdir, '-lctdet', '-Wl,-rpath,' + libdir, '-o', exe])
    r = subprocess.run([exe, 'gpu'], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert '_nms contract keeps 2: 0 2' in r.stdout


def test_cpp_caller_runs_the_reference_symbol_on_the_device(tmp_path):
    """examples/cpp_caller.cpp knows only the reference's prototype (utils/nms/gpu_nms.hpp:1-2, C++ linkage) and gets
    the known answer from libctdet's `_nms`."""
    import os
    import shutil
    import subprocess
    if shutil.which('g++') is None:
        pytest.skip('g++ not available')
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(repo, 'context-transformer_amd', 'lib')
    exe = str(tmp_path / 'cpp_caller')
    subprocess.check_call(['g++', os.path.join(repo, 'examples', 'cpp_caller.cpp'), '-L' + libdir, '-lctdet',
                           '-Wl,-rpath,' + libdir, '-o', exe])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert '_nms keeps 2: 0 2' in r.stdout
