"""Pin oracle/ (the CPU restatement) against golden vectors captured from the reference
itself (tools/gen_goldens.py).  CPU only."""
import hashlib

import numpy as np
import pytest
import torch

from conftest import rel_err, sampled
from ctdet import synth
from oracle import box_ref, loss_ref, nms_ref, rfbnet_ref

torch.set_num_threads(8)
TOL = 1e-4          # fp32 activations: max|a-b|/max|b|  (north_star tolerance)


# ---------------------------------------------------------------- boxes
@pytest.mark.parametrize('name', sorted(box_ref.ANCHOR_CFGS))
def test_prior_box_bit_exact(golden, name):
    g = golden('box_ops.npz')
    p = box_ref.prior_box(box_ref.ANCHOR_CFGS[name]).numpy()
    assert list(p.shape) == list(g['prior_%s__shape' % name])
    sha = np.frombuffer(hashlib.sha256(p.tobytes()).digest(), dtype=np.uint8)
    assert (sha == g['prior_%s__sha' % name]).all()
    assert np.array_equal(p[::97], g['prior_%s__rows' % name])


def test_prior_box_known_answers():
    # SURVEY 8c anchors
    p = box_ref.prior_box(box_ref.ANCHOR_CFGS['VOC_300']).numpy()
    assert p.shape == (11620, 4)
    assert abs(float(p.astype(np.float64).sum()) - 15500.051000) < 1e-3
    assert hashlib.sha256(p.tobytes()).hexdigest()[:16] == '919d5cbdd009a845'
    np.testing.assert_allclose(p[5], [0.013333, 0.013333, 0.057735, 0.173205], atol=1e-6)
    p5 = box_ref.prior_box(box_ref.ANCHOR_CFGS['VOC_512']).numpy()
    assert p5.shape == (32756, 4)
    assert hashlib.sha256(p5.tobytes()).hexdigest()[:16] == 'a4ac9c18513efaf8'


def test_box_coding_bit_exact(golden):
    g = golden('box_ops.npz')
    priors = box_ref.prior_box(box_ref.ANCHOR_CFGS['VOC_300'])
    gen = torch.Generator().manual_seed(7)
    loc = torch.randn(priors.shape[0], 4, generator=gen)
    assert np.array_equal(box_ref.decode(loc, priors, [0.1, 0.2]).numpy()[::7], g['decode_out'])
    assert np.array_equal(box_ref.point_form(priors).numpy()[::7], g['point_form_out'])
    truths = torch.from_numpy(g['match_truths'])
    labels = torch.from_numpy(g['match_labels'])
    ov = box_ref.jaccard(truths, box_ref.point_form(priors))
    assert np.array_equal(ov.numpy()[:, ::5], g['jaccard_out'])
    assert np.array_equal(ov.max(0)[1].numpy(), g['jaccard_argmax0'])
    matched = truths[ov.max(0)[1]]
    assert np.array_equal(box_ref.encode(matched, priors, [0.1, 0.2]).numpy()[::7], g['encode_out'], equal_nan=True)
    for thr in (0.5, 0.35):
        tag = 'match%02d' % int(thr * 100)
        loc_t, conf_t, obj_t, ovl = box_ref.match(thr, truths, priors, [0.1, 0.2], labels)
        assert np.array_equal(loc_t.numpy()[::7], g[tag + '_loc'], equal_nan=True)
        assert np.array_equal(conf_t.numpy(), g[tag + '_conf'])
        assert np.array_equal(obj_t.numpy(), g[tag + '_obj'])
        assert np.array_equal(ovl.numpy()[::7], g[tag + '_overlap'])
    # known answers
    ka = box_ref.decode(torch.tensor([[1, -2, .5, -.5], [0, 0, 0, 0.]]),
                        torch.tensor([[.5, .5, .2, .4], [.1, .9, .3, .3]]), [0.1, 0.2]).numpy()
    assert np.array_equal(ka, g['ka_decode'])
    np.testing.assert_allclose(ka[0], [0.4094828963, 0.2390324920, 0.6305171251, 0.6009674668], rtol=1e-6)
    kj = box_ref.jaccard(torch.tensor([[0, 0, .5, .5]]),
                         torch.tensor([[.25, .25, .75, .75], [0, 0, .5, .5], [.6, .6, .9, .9]])).numpy()
    assert np.array_equal(kj, g['ka_jaccard'])
    ke = box_ref.encode(torch.tensor([[.3, .3, .7, .8]]), torch.tensor([[.5, .5, .2, .4]]), [0.1, 0.2]).numpy()
    assert np.array_equal(ke, g['ka_encode'])


def test_detect_and_box_utils_nms(golden):
    g = golden('box_ops.npz')
    priors = box_ref.prior_box(box_ref.ANCHOR_CFGS['VOC_300'])
    P = priors.shape[0]
    gen = torch.Generator().manual_seed(11)
    loc = torch.randn(2, P, 4, generator=gen)
    conf = torch.softmax(torch.randn(2, P, 20, generator=gen) * 2, -1)
    obj = torch.softmax(torch.randn(2, P, 2, generator=gen), -1)
    boxes, scores = box_ref.detect(loc, conf, obj, priors)
    for t, name in ((boxes, 'detect_boxes'), (scores, 'detect_scores')):
        a, b, s, gs = sampled(t, g, name)
        assert np.array_equal(a, b)
    sc = torch.rand(P, generator=gen)
    for (ovt, topk) in ((0.5, 200), (0.3, 400)):
        keep, count = box_ref.box_utils_nms(boxes[0], sc, ovt, topk)
        assert np.array_equal(keep[:count].numpy(), g['bunms_%02d_%d_keep' % (int(ovt * 100), topk)])


# ---------------------------------------------------------------- NMS
def test_nms_oracle_vs_reference(golden):
    g = golden('nms.npz')
    have_ge = bool(g['have_ge'])
    for ci in range(int(g['ncases'])):
        d = g['c%d_dets' % ci]
        assert len(np.unique(d[:, 4])) == len(d), 'golden must be tie-free'
        for thr in (0.45, 0.3, 0.5, 0.7):
            tag = 'c%d_t%02d' % (ci, int(round(thr * 100)))
            assert np.array_equal(nms_ref.nms(d, thr, ge=False), g[tag + '_gt']), tag
            assert np.array_equal(nms_ref.nms_c(d, thr, ge=False), g[tag + '_gt']), tag
            if have_ge:
                assert np.array_equal(nms_ref.nms(d, thr, ge=True), g[tag + '_ge']), tag
                assert np.array_equal(nms_ref.nms_c(d, thr, ge=True), g[tag + '_ge']), tag
    assert list(nms_ref.nms(g['ka_dets'], 0.45)) == [0, 2] == list(g['ka_gt'])
    # IoU exactly == thresh: '>' keeps both, '>=' suppresses
    assert list(nms_ref.nms(g['eq_dets'], 0.5, ge=False)) == [0, 1] == list(g['eq_gt'])
    assert list(nms_ref.nms(g['eq_dets'], 0.5, ge=True)) == [0]
    if have_ge:
        assert list(g['eq_ge']) == [0]


def test_soft_nms_oracle(golden):
    g = golden('nms.npz')
    if not bool(g['have_ge']):
        pytest.skip('no patched reference build was available when goldens were made')
    for m in (0, 1, 2):
        b = g['c6_dets'].copy()
        n = nms_ref.soft_nms(b, 0.5, 0.3, 0.001, m)
        assert n == int(g['soft_m%d_n' % m])
        ref = g['soft_m%d_boxes' % m]
        assert np.array_equal(b[:n, :4], ref[:n, :4])
        np.testing.assert_allclose(b[:n, 4], ref[:n, 4], rtol=2e-6, atol=0)


def test_pipeline_oracle(golden):
    g = golden('pipeline.npz')
    priors = box_ref.prior_box(box_ref.ANCHOR_CFGS['VOC_300'])
    P = priors.shape[0]
    gen = torch.Generator().manual_seed(int(g['seed']))
    B, T = 2, 20
    loc = torch.randn(B, P, 4, generator=gen) * 0.5
    conf = torch.softmax(torch.randn(B, P, T, generator=gen) * 3.0, -1)
    obj = torch.softmax(torch.randn(B, P, 2, generator=gen) * 2.0 + torch.tensor([2.5, 0.0]), -1)
    boxes, scores = box_ref.detect(loc, conf, obj, priors)
    for i in range(B):
        out = nms_ref.postprocess_image(boxes[i].numpy(), scores[i].numpy(), (500, 375), nms_fn=nms_ref.nms_c)
        for j in range(1, T + 1):
            assert np.array_equal(out[j], g['img%d_cls%d' % (i, j)]), (i, j)


# ---------------------------------------------------------------- model
def _check(t, g, name, tol=TOL):
    a, b, s, gs = sampled(t, g, name)
    e = rel_err(a, b)
    assert e <= tol, (name, e)


def test_state_dict_contract(golden):
    for fname, (size, C, phase, setting) in {
            'rfb300_phase1.npz': (300, 20, 1, 'transfer'),
            'rfb300_phase2_transfer.npz': (300, 60, 2, 'transfer'),
            'rfb300_phase2_incre.npz': (300, 15, 2, 'incre'),
            'rfb512_phase1.npz': (512, 20, 1, 'transfer')}.items():
        g = golden(fname)
        shapes = rfbnet_ref.param_shapes(size, C, phase, 'ours', setting)
        assert sorted(shapes) == sorted(g['keys'].tolist()), fname
        n = sum(int(np.prod(s)) for k, s in shapes.items()
                if not k.endswith(('running_mean', 'running_var', 'num_batches_tracked')))
        assert n == int(g['nparams']), fname
    assert int(golden('rfb300_phase1.npz')['nparams']) == 36674176
    assert int(golden('rfb300_phase2_transfer.npz')['nparams']) == 42401617
    assert int(golden('rfb512_phase1.npz')['nparams']) == 37574620
    assert len(golden('rfb300_phase2_transfer.npz')['keys']) == 381


def test_rfb300_phase1(golden):
    g = golden('rfb300_phase1.npz')
    sd = synth.fill_state_dict(rfbnet_ref.param_shapes(300, 20, 1))
    x = synth.images(2, 300, 'randn', 1234)
    with torch.no_grad():
        srcs = rfbnet_ref.backbone(sd, x, 300)
        _check(srcs[0], g, 'p1_norm')
        for k, name in ((1, 'extra0'), (2, 'extra1'), (3, 'extra2'), (4, 'extra4'), (5, 'extra6')):
            _check(srcs[k], g, 'p1_' + name)
        loc, conf, obj = rfbnet_ref.forward(sd, x, 300, 20)
        _check(loc, g, 'p1_loc'); _check(conf, g, 'p1_conf'); _check(obj, g, 'p1_obj')
        _check(rfbnet_ref.forward(sd, x, 300, 20, init=True), g, 'p1_init_conf')
        priors = box_ref.prior_box(box_ref.ANCHOR_CFGS['VOC_300'])
        boxes, scores = box_ref.detect(loc, conf, obj, priors)
        _check(boxes, g, 'p1_boxes'); _check(scores, g, 'p1_scores')
        loc, conf, obj = rfbnet_ref.forward(sd, synth.images(1, 300, 'u8', 1234), 300, 20)
        _check(loc, g, 'p1u8_loc'); _check(conf, g, 'p1u8_conf'); _check(obj, g, 'p1u8_obj')
        loc, conf, obj = rfbnet_ref.forward(sd, x, 300, 20, training=True)
        _check(loc, g, 'p1tr_loc'); _check(conf, g, 'p1tr_conf'); _check(obj, g, 'p1tr_obj')


@pytest.mark.parametrize('setting,C', [('transfer', 60), ('incre', 15)])
def test_rfb300_phase2(golden, setting, C):
    g = golden('rfb300_phase2_%s.npz' % setting)
    sd = synth.fill_state_dict(rfbnet_ref.param_shapes(300, C, 2, 'ours', setting))
    x = synth.images(2, 300, 'randn', 1234)
    with torch.no_grad():
        loc, conf, obj = rfbnet_ref.forward(sd, x, 300, C, 2, 'ours', setting)
        _check(loc, g, 'loc'); _check(conf, g, 'conf'); _check(obj, g, 'obj')
        _check(rfbnet_ref.forward(sd, x, 300, C, 2, 'ours', setting, init=True), g, 'init_conf')
        loc, conf, obj = rfbnet_ref.forward(sd, x, 300, C, 2, 'ours', setting, training=True)
        _check(loc, g, 'tr_loc'); _check(conf, g, 'tr_conf'); _check(obj, g, 'tr_obj')


def test_rfb512_phase1(golden):
    g = golden('rfb512_phase1.npz')
    sd = synth.fill_state_dict(rfbnet_ref.param_shapes(512, 20, 1))
    x = synth.images(1, 512, 'randn', 1234)
    with torch.no_grad():
        loc, conf, obj = rfbnet_ref.forward(sd, x, 512, 20)
    assert loc.shape == (1, 32756, 4)
    _check(loc, g, 'p1_loc'); _check(conf, g, 'p1_conf'); _check(obj, g, 'p1_obj')


def test_loss(golden):
    g = golden('loss.npz')
    priors = box_ref.prior_box(box_ref.ANCHOR_CFGS['VOC_300'])
    for tag, (C, phase, ncls) in {'p1': (20, 1, 21), 'p2': (60, 2, 21)}.items():
        sd = synth.fill_state_dict(rfbnet_ref.param_shapes(300, C, phase))
        x = synth.images(2, 300, 'randn', 1234)
        tg = [torch.from_numpy(g['%s_target%d' % (tag, i)]) for i in range(2)]
        with torch.no_grad():
            out = rfbnet_ref.forward(sd, x, 300, C, phase, training=True)
        for name, o in zip(('loc', 'conf', 'obj'), out):
            _check(o, g, '%s_in_%s' % (tag, name))
        out = tuple(o.detach().requires_grad_(True) for o in out)
        ld = loss_ref.multibox_loss_combined(out, priors, tg, ncls)
        sum(ld.values()).backward()
        for k, v in ld.items():
            assert abs(v.item() - float(g['%s_%s' % (tag, k)])) <= 2e-4 * max(1.0, abs(float(g['%s_%s' % (tag, k)]))), (tag, k)
        for name, o in zip(('loc', 'conf', 'obj'), out):
            _check(o.grad, g, '%s_grad_%s' % (tag, name), tol=2e-3)
