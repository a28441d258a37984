"""End-to-end parity on the MI355X: the full RFBNet engine (HIP convs / pools / attention) and the
batched detection pipeline against the CPU oracle and the committed goldens."""
import types

import numpy as np
import pytest
import torch

from conftest import rel_err, sampled
from ctdet import ops, synth
from ctdet.pipeline import DetectionPipeline
from oracle import box_ref, nms_ref, rfbnet_ref

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TOL = 1e-4


def _net(size, C, phase=1, setting='transfer'):
    from models.RFB_Net_vgg import build_net
    args = types.SimpleNamespace(method='ours', phase=phase, setting=setting)
    net = build_net(args, size, C)
    net.load_state_dict(synth.fill_state_dict(net.state_dict()), strict=True)
    net = net.eval().cuda()
    net.device = 'cuda'
    return net


def _layer_report(net, x, size):
    """Per-source error table (printed when a model-level assertion fails)."""
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    with torch.no_grad():
        srcs = rfbnet_ref.backbone(sd, x, size)
    rt = net.runtime(x.shape[0])
    names = ['Norm.out'] + ['extras.%d.out' % k if ('extras.%d.out' % k) in rt.bufs else 'a_extras.%d' % k
                            for k in range(len(net.extras)) if k < net.indicator or k % 2 == 0]
    rows = []
    for nme, s in zip(names, srcs):
        rows.append('%s: %.3e' % (nme, rel_err(rt.bufs[nme].cpu(), s)))
    return '; '.join(rows)


def test_rfb300_phase1_vs_oracle_and_golden(golden):
    g = golden('rfb300_phase1.npz')
    net = _net(300, 20)
    x = synth.images(2, 300, 'randn', 1234)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    with torch.no_grad():
        want_raw = rfbnet_ref.forward(sd, x, 300, 20, raw=True)
        got_raw = [t.cpu() for t in net.forward_raw(x.cuda())]
        errs = [rel_err(a, b) for a, b in zip(got_raw, want_raw)]
        assert max(errs) < TOL, (errs, _layer_report(net, x, 300))
        loc, conf, obj = [t.cpu() for t in net(x)]
    for t, name in ((loc, 'p1_loc'), (conf, 'p1_conf'), (obj, 'p1_obj')):
        a, b, _, _ = sampled(t, g, name)
        assert rel_err(a, b) < TOL, name
    with torch.no_grad():
        init_conf = net(x, init=True).cpu()
    a, b, _, _ = sampled(init_conf, g, 'p1_init_conf')
    assert rel_err(a, b) < TOL
    # Detect on the device outputs (reference call sequence test.py:130-131)
    from layers.functions import Detect, PriorBox
    from data import VOC_300
    priors = PriorBox(VOC_300).forward().cuda()
    boxes, scores = Detect(21, 0, VOC_300).forward(net(x), priors)
    for t, name in ((boxes.cpu(), 'p1_boxes'), (scores.cpu(), 'p1_scores')):
        a, b, _, _ = sampled(t, g, name)
        assert rel_err(a, b) < TOL, name
    # image-like input
    xu = synth.images(1, 300, 'u8', 1234)
    with torch.no_grad():
        want_raw = rfbnet_ref.forward(sd, xu, 300, 20, raw=True)
        got_raw = [t.cpu() for t in net.forward_raw(xu.cuda())]
        assert max(rel_err(a, b) for a, b in zip(got_raw, want_raw)) < TOL      # activations: 1e-4
        loc, conf, obj = [t.cpu() for t in net(xu)]
    # 0..255 pixel inputs give |logit| ~ 1e2; softmax turns a 1e-5 relative logit error into
    # up to ~|logit|*1e-5 in the probabilities, so the post-softmax goldens get 2e-3
    for t, name in ((loc, 'p1u8_loc'), (conf, 'p1u8_conf'), (obj, 'p1u8_obj')):
        a, b, _, _ = sampled(t, g, name)
        assert rel_err(a, b) < (TOL if name.endswith('loc') else 2e-3), name


@pytest.mark.parametrize('size,batch', [(300, 2), (512, 4)])
def test_f16x2_operand_forms_whole_network(monkeypatch, size, batch):
    """The whole network on the f16x2 operand forms (csrc/ct_f16x2.h; CTDET_H2=2 forces what batch x size^2 >= 8 x 300^2 selects by
    default, engine.operand_form_h2): raw outputs against the CPU oracle at 1e-4, the per-image maxima wired from producer to
    consumer (no absmax launches of their own), and an image's outputs independent of its batch mates, bit for bit."""
    from ctdet import engine
    monkeypatch.setenv('CTDET_H2', '2')
    net = _net(size, 20)
    x = synth.images(batch, size, 'randn', 4321)
    x[batch - 1] *= 37.0                                   # a batch mate with other maxima
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    with torch.no_grad():
        want = rfbnet_ref.forward(sd, x, size, 20, raw=True)
        got = [t.cpu().clone() for t in net.forward_raw(x.cuda())]
    rt = net.runtime(batch)
    assert rt.backend.h2
    kinds = [st.rt.get('wino') for st in rt.conv_steps()]
    assert any(k in engine.H2_TILES for k in kinds), kinds
    assert not any(k == 46 for k in kinds) and rt.amax_slots, kinds       # every f16x2 consumer got its producer's maxima
    assert all(st.rt.get('amax_own') is None for st in rt.conv_steps())
    for g_, w_ in zip(got, want):
        for n in range(batch):
            assert rel_err(g_[n], w_[n]) < TOL, n
    x2 = x.clone()
    x2[batch - 1] = synth.images(1, size, 'randn', 99)[0] * 0.01          # another mate, other maxima again
    with torch.no_grad():
        other = [t.cpu() for t in net.forward_raw(x2.cuda())]
    for a_, g_ in zip(other, got):
        assert torch.equal(a_[:batch - 1], g_[:batch - 1])


@pytest.mark.parametrize('setting,C', [('transfer', 60), ('incre', 15)])
def test_rfb300_phase2_context_transformer(golden, setting, C):
    g = golden('rfb300_phase2_%s.npz' % setting)
    net = _net(300, C, 2, setting)
    x = synth.images(2, 300, 'randn', 1234)
    with torch.no_grad():
        loc, conf, obj = [t.cpu() for t in net(x)]
        init_conf = net(x, init=True).cpu()
    for t, name in ((loc, 'loc'), (conf, 'conf'), (obj, 'obj'), (init_conf, 'init_conf')):
        a, b, _, _ = sampled(t, g, name)
        assert rel_err(a, b) < TOL, (name, rel_err(a, b))
    assert conf.shape[-1] == (20 if setting == 'transfer' else 20)


def test_winograd_tile_policy_and_phase2_full_tensor_parity(monkeypatch):
    """Networks with the Context-Transformer block (whose softmax amplifies the trunk's fp32 rounding ~1000x) have a measured
    kernel policy (engine.ctx_policy / ctx_tile_set / operand_form_h2).  Shipped (round 6, 'h2'): the committed table with its
    F(4x4,3x3) entries on the f16x2 operand form at every batch size, the direct layers on bf16x3.  CTDET_CTX_TILES=2,23 is round
    5's tile-set policy: F(2x2,3x3) / bf16x3 with two accumulators (tile 23), a fused F(4x4,3x3) kernel only up to 128 input
    channels (ctx_f4_max_cin, ctx_f4_tile), the three-kernel form opt-in (ctx_w4s_min_cin); 'any' = what every other network
    runs.  Every output ELEMENT (not a sample) of the block stays within 1e-4 of the reference's CPU arithmetic here (bs 2, seed
    1234; tests/test_gpu_ctx_parity.py sweeps batch sizes and seeds)."""
    from ctdet import engine
    net = _net(300, 60, 2, 'transfer')
    rt = net.runtime(2)
    for r in (rt, net.runtime(32)):
        assert r.backend.h2 and not r.backend.h2_direct and r.backend.wino_tile_set is None
        tiles = [st.rt.get('wino') for st in r.conv_steps() if st.rt.get('wino')]
        assert tiles and not set(tiles) & {44, 45, 46}, tiles             # every bf16x3 F(4x4) entry runs its f16x2 twin
        assert not any(st.rt.get('x3') is not None and r.backend.x3_h2(st.rt['x3']) for st in r.conv_steps())
    assert any(st.rt.get('wino') == 47 and st.cin == 512 for st in net.runtime(32).conv_steps())
    x = synth.images(2, 300, 'randn', 1234)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    with torch.no_grad():
        want = rfbnet_ref.forward(sd, x, 300, 60, 2, 'ours', 'transfer', raw=True)
        got = [t.cpu() for t in net.forward_raw(x.cuda())]
    for a, b, name in zip(got, want, ('loc', 'conf', 'obj')):
        assert rel_err(a.reshape(b.shape), b) < TOL, (name, rel_err(a.reshape(b.shape), b))
    # round 5's tile-set policy
    monkeypatch.setenv('CTDET_CTX_TILES', '2,23')
    old = _net(300, 60, 2, 'transfer')
    for r in (old.runtime(2), old.runtime(32)):
        assert not r.backend.h2
        tiles = [st.rt.get('wino') for st in r.conv_steps() if st.rt.get('wino')]
        assert tiles and set(tiles) <= {2, 4, 23, 46}, tiles
        assert all(st.rt.get('wino') == 23 for st in r.conv_steps() if st.rt.get('wino') and st.cin % 16 == 0 and st.cin > 128)
        assert all(st.cin <= 128 for st in r.conv_steps() if st.rt.get('wino') in (4, 46))
    assert any(st.rt.get('wino') == 23 and st.cin == 512 for st in old.runtime(32).conv_steps())
    with torch.no_grad():
        got = [t.cpu() for t in old.forward_raw(x.cuda())]
    for a, b, name in zip(got, want, ('loc', 'conf', 'obj')):
        assert rel_err(a.reshape(b.shape), b) < TOL, (name, rel_err(a.reshape(b.shape), b))
    del old
    # ... and its opt-in fast variant of round 4: three-kernel F(4x4) from 128 input channels up
    monkeypatch.setenv('CTDET_CTX_W4S_MIN_CIN', '128')
    monkeypatch.setenv('CTDET_CTX_F4_MAX_CIN', '128')
    fast = _net(300, 60, 2, 'transfer').runtime(32)
    assert any(st.rt.get('wino') == 44 and st.cin == 512 for st in fast.conv_steps())
    del fast
    monkeypatch.setenv('CTDET_CTX_F4_MAX_CIN', '0')
    monkeypatch.setenv('CTDET_CTX_W4S_MIN_CIN', '0')
    monkeypatch.setenv('CTDET_CTX_TILES', '2')
    assert {st.rt.get('wino') for st in _net(300, 60, 2, 'transfer').runtime(32).conv_steps() if st.rt.get('wino')} == {2}
    monkeypatch.setenv('CTDET_CTX_TILES', 'any')
    free = _net(300, 60, 2, 'transfer').runtime(32)
    plain = _net(300, 20).runtime(32)
    assert sorted(st.rt.get('wino') or 0 for st in free.conv_steps() if not st.segs) == \
        sorted(st.rt.get('wino') or 0 for st in plain.conv_steps() if not st.segs)


def test_rfb512_phase1(golden):
    g = golden('rfb512_phase1.npz')
    net = _net(512, 20)
    x = synth.images(1, 512, 'randn', 1234)
    with torch.no_grad():
        loc, conf, obj = [t.cpu() for t in net(x)]
    assert loc.shape == (1, 32756, 4)
    for t, name in ((loc, 'p1_loc'), (conf, 'p1_conf'), (obj, 'p1_obj')):
        a, b, _, _ = sampled(t, g, name)
        assert rel_err(a, b) < TOL, (name, rel_err(a, b), _layer_report(net, x, 512))


def test_rfb512_context_transformer_build_defined():
    """parity unpinned: the reference raises IndexError at 512 (models/RFB_Net_vgg.py:235-244);
    the build's 7-entry pooling list is checked against the build's own CPU restatement only."""
    net = _net(512, 60, 2, 'transfer')
    x = synth.images(1, 512, 'randn', 1234)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    with torch.no_grad():
        want = rfbnet_ref.forward(sd, x, 512, 60, 2, 'ours', 'transfer', raw=True)
        want64 = rfbnet_ref.forward(sd64, x.double(), 512, 60, 2, 'ours', 'transfer', raw=True)
        got = [t.cpu() for t in net.forward_raw(x.cuda())]
    assert net.runtime(1).plan.M == 4964
    # The un-scaled theta.phi^T logits reach |x| ~ 1e2-1e3 here, so fp32 itself (CPU or GPU) is
    # only good to ~1e-3 after the softmax; judge both fp32 paths against an fp64 evaluation.
    for name, a, b, c in zip(('loc', 'conf', 'obj'), got, want, want64):
        e_gpu, e_cpu = rel_err(a, c), rel_err(b, c)
        assert e_gpu < max(TOL, 5 * e_cpu), (name, e_gpu, e_cpu)


def test_batched_pipeline_equals_sequential_reference_loop():
    """DetectionPipeline (bs=4) == the reference's per-image / per-class loop (test.py:121-161)
    applied to the SAME device-produced boxes/scores: bit-exact rows; and the boxes/scores
    themselves match the CPU oracle within the fp32 tolerance."""
    from layers.functions import PriorBox
    from data import VOC_300
    from utils.nms_wrapper import nms
    B, T = 4, 20
    net = _net(300, 20)
    priors = PriorBox(VOC_300).forward()
    x = synth.images(B, 300, 'randn', 1234)
    pipe = DetectionPipeline(net, priors, B, T, image_wh=(500, 375))
    pipe.run(x.cuda())
    got = pipe.results()
    boxes, scores = pipe.boxes.cpu().numpy(), pipe.scores.cpu().numpy()
    ncand = 0
    for i in range(B):
        want = nms_ref.postprocess_image(boxes[i], scores[i], (1, 1), nms_fn=nms_ref.nms_c)
        for j in range(1, T + 1):
            ncand += int((scores[i, :, j] > 0.01).sum())
            assert np.array_equal(got[i][j], want[j]), (i, j, got[i][j].shape, want[j].shape)
    assert ncand > 1000, 'degenerate case: no candidates'
    # the reference call sequence on one image through the drop-in `nms` (device kernel, '>' rule)
    i = 1
    per_cls = []
    for j in range(1, T + 1):
        inds = np.where(scores[i, :, j] > 0.01)[0]
        c_dets = np.hstack((boxes[i][inds], scores[i][inds, j][:, None])).astype(np.float32)
        keep = nms(c_dets, 0.45) if len(inds) else []
        per_cls.append(c_dets[keep, :] if len(inds) else np.empty((0, 5), np.float32))
    sc = np.hstack([d[:, -1] for d in per_cls])
    if len(sc) > 200:
        th = np.sort(sc)[-200]
        per_cls = [d[d[:, -1] >= th] for d in per_cls]
    for j in range(T):
        assert np.array_equal(per_cls[j], got[i][j + 1]), j
    # numerics of the fused softmax/decode stage against the oracle
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    with torch.no_grad():
        loc, conf, obj = rfbnet_ref.forward(sd, x, 300, 20)
    wb, ws = box_ref.detect(loc, conf, obj, priors)
    assert rel_err(boxes, (wb * torch.tensor([500., 375., 500., 375.])).numpy()) < TOL
    assert rel_err(scores, ws.numpy()) < TOL


def test_weights_repacked_after_update():
    net = _net(300, 20)
    x = synth.images(1, 300, 'randn', 7).cuda()
    with torch.no_grad():
        a = net.forward_raw(x)[1].clone()
        net.conf[0].bias.add_(0.5)
        b = net.forward_raw(x)[1].clone()
    d = (b - a).view(-1)[:38 * 38 * 6 * 20]
    assert torch.allclose(d, torch.full_like(d, 0.5), atol=1e-4)


def test_large_batch_512_matches_small_batches():
    """bs=32 at 512x512 makes base.2's input 2.1 GB, i.e. more than one buffer descriptor can
    address: ct_conv2d_fwd splits the batch.  Every image must come out exactly as it does in a
    bs=4 run of the same engine (both use the same tile configs only if tuned alike, so compare
    within the fp32 tolerance)."""
    net = _net(512, 20)
    x = synth.images(32, 512, 'randn', 99)
    with torch.no_grad():
        big = [t.clone() for t in net.forward_raw(x.cuda())]
        for i0 in (0, 12, 28):
            small = net.forward_raw(x[i0:i0 + 4].cuda())
            for a, b in zip(big, small):
                assert rel_err(a[i0:i0 + 4].cpu(), b.cpu()) < 1e-5
    del net
    torch.cuda.empty_cache()


def test_schedule_variants_agree(monkeypatch):
    """The execution variants of the engine are pure scheduling / algebra choices: one stream vs the two-stream
    schedule and fused vs separate MaxPool2d(2,2) give bit-identical outputs, Winograd vs direct convolution
    agrees to the parity tolerance."""
    x = synth.images(4, 300, 'randn', 77).cuda()

    def run(**env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        net = _net(300, 20)
        out = [t.clone() for t in net.forward_raw(x)]
        rt = net.runtime(4)
        for k in env:
            monkeypatch.delenv(k)
        return out, rt

    base, rt = run()
    assert rt.side is not None and any(st.rt.get('wino') for st in rt.conv_steps())
    assert any(getattr(st, 'fused_into', None) for st in rt.plan.steps)
    one, rt1 = run(CTDET_STREAMS='1')
    assert rt1.side is None
    for a, b in zip(base, one):
        assert torch.equal(a, b)
    sep, rt2 = run(CTDET_FUSE_POOL='0')
    assert not any(getattr(st, 'fused_into', None) for st in rt2.plan.steps)
    for a, b in zip(base, sep):
        assert torch.equal(a, b)
    direct, rt3 = run(CTDET_WINO='0', CTDET_TUNE='0')
    assert not any(st.rt.get('wino') for st in rt3.conv_steps())
    for a, b in zip(base, direct):
        assert rel_err(a.cpu(), b.cpu()) < TOL


def test_full_size_pipeline_properties():
    """BASELINE configs[1] at full size (RFBNet-300, bs 32, 20 classes): properties that do not need the oracle
    at that size -- batch-position invariance (exact), descending scores, the top-200 rule, NMS idempotence."""
    from layers.functions import PriorBox
    from data import VOC_300
    from utils.nms_wrapper import nms
    net = _net(300, 20)
    priors = PriorBox(VOC_300).forward()
    pipe = DetectionPipeline(net, priors, 32, 20)
    x = synth.images(32, 300, 'randn', 2024).cuda()
    pipe.run(x)
    base = pipe.results()
    perm = torch.randperm(32, generator=torch.Generator().manual_seed(3))
    pipe.run(x[perm.cuda()].contiguous())
    shuffled = pipe.results()
    for new_pos, old_pos in enumerate(perm.tolist()):
        for j in range(1, 21):
            assert np.array_equal(shuffled[new_pos][j], base[old_pos][j]), (new_pos, old_pos, j)
    total = 0
    for i in range(32):
        sc = np.concatenate([base[i][j][:, 4] for j in range(1, 21)])
        total += len(sc)
        assert len(sc) > 0 and np.isfinite(np.concatenate([base[i][j] for j in range(1, 21)])).all()
        if len(sc) > 200:
            assert np.sum(sc > sc.min()) < 200                 # only ties at the 200-th score may exceed it
        for j in range(1, 21):
            d = base[i][j]
            assert np.all(np.diff(d[:, 4]) <= 0)
            if len(d) > 1 and i % 8 == 0:                        # kept boxes do not suppress each other
                assert list(nms(d, 0.45)) == list(range(len(d)))
    assert total >= 32 * 150


def test_full_size_pipeline_vs_oracle_on_first_and_last_image():
    """BASELINE configs[1] at FULL size (RFBNet-300, bs 32) against the oracle: the CPU oracle evaluates images 0 and
    31 of the same batch (the kernels are batch-position invariant, test above); raw loc / conf / obj within 1e-4,
    and the detections of the batched device post-processing bit-exact against the reference's sequential per-class
    loop (test.py:136-161, oracle/nms_ref.py with the C NMS) run on the device's own boxes / scores."""
    from layers.functions import PriorBox
    from data import VOC_300
    net = _net(300, 20)
    priors = PriorBox(VOC_300).forward()
    pipe = DetectionPipeline(net, priors, 32, 20, image_wh=(500, 375))
    x = synth.images(32, 300, 'randn', 2024)
    pipe.run(x.cuda())
    got = pipe.results()
    idx = [0, 31]
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    with torch.no_grad():
        want = rfbnet_ref.forward(sd, x[idx], 300, 20, raw=True)
        raw = [t.cpu()[idx] for t in net.forward_raw(x.cuda())]
    for a, b, name in zip(raw, want, ('loc', 'conf', 'obj')):
        assert rel_err(a.reshape(b.shape), b) < TOL, (name, rel_err(a.reshape(b.shape), b))
    boxes, scores = pipe.boxes.cpu().numpy(), pipe.scores.cpu().numpy()
    nms_ref.build_c()
    for i in idx:
        ref = nms_ref.postprocess_image(boxes[i], scores[i], (1, 1), nms_fn=nms_ref.nms_c)
        for j in range(1, 21):
            assert np.array_equal(got[i][j], ref[j]), (i, j)


def test_full_size_512_pipeline_vs_oracle_on_first_and_last_image():
    """north_star quotes its targets "at 512x512 bs=32", and the bench times that shape (other_configs.rfb512_bs32): RFBNet-512,
    bs 32, 20 classes against the oracle on images 0 and 31 of the batch -- raw loc / conf / obj within 1e-4 of the CPU path
    (oracle/rfbnet_ref.py), detections of the batched device post-processing bit-exact against the reference's sequential
    per-class loop (test.py:136-161, oracle/nms_ref.py with the C NMS) on the device's own boxes / scores.  The tile choices at
    this shape (conv1_2 .. conv3_3 on 512 x 512 .. 128 x 128 maps, the >2 GiB batch split of conv1_x) are exercised nowhere else
    against the oracle; test_large_batch_512_matches_small_batches only compares the engine with itself.
    models/RFB_Net_vgg.py:190-286."""
    from layers.functions import PriorBox
    from data import VOC_512
    net = _net(512, 20)
    priors = PriorBox(VOC_512).forward()
    pipe = DetectionPipeline(net, priors, 32, 20, image_wh=(500, 375))
    x = synth.images(32, 512, 'randn', 2025)
    pipe.run(x.cuda())
    got = pipe.results()
    idx = [0, 31]
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    with torch.no_grad():
        want = rfbnet_ref.forward(sd, x[idx], 512, 20, raw=True)
        raw = [t.cpu()[idx] for t in net.forward_raw(x.cuda())]
    for a, b, name in zip(raw, want, ('loc', 'conf', 'obj')):
        assert rel_err(a.reshape(b.shape), b) < TOL, (name, rel_err(a.reshape(b.shape), b))
    boxes, scores = pipe.boxes.cpu().numpy(), pipe.scores.cpu().numpy()
    nms_ref.build_c()
    ncand = 0
    for i in idx:
        ref = nms_ref.postprocess_image(boxes[i], scores[i], (1, 1), nms_fn=nms_ref.nms_c)
        for j in range(1, 21):
            ncand += len(ref[j])
            assert np.array_equal(got[i][j], ref[j]), (i, j)
    assert ncand > 100, 'degenerate case: no detections'
    del pipe, net
    torch.cuda.empty_cache()


def test_hipgraph_replay_equals_eager_launches(monkeypatch):
    """DetectionPipeline captures its step (two streams, ~100 launches) into a hipGraph after two eager steps; the
    replays give bit-identical detections, keep doing so after a weight update (re-packing happens outside the
    graph into the same buffers), a per-image scale change and a new input -- and really are replays."""
    from layers.functions import PriorBox
    from data import VOC_300
    monkeypatch.setenv('CTDET_TUNE', '0')       # shapes missing from the committed table would be timed live: the two
    priors = PriorBox(VOC_300).forward()        # pipelines below could then pick different tiles (other rounding)
    B = 4
    xs = [synth.images(B, 300, 'randn', 100 + i).cuda() for i in range(3)]

    def collect(graph):
        net = _net(300, 60, 2, 'transfer')           # with the Context-Transformer block on the path
        pipe = DetectionPipeline(net, priors, B, 20, graph=graph)
        out = []
        for it in range(6):
            if it == 3:
                with torch.no_grad():
                    net.conf[1].bias.add_(0.25)
                    net.Wz.mul_(1.5)
            wh = torch.tensor([[500., 375.], [640., 480.], [300. + it, 300.], [1000., 200.]]) if it >= 4 else None
            pipe.run(xs[it % 3], image_wh=wh)
            out.append(([[d.copy() for d in img] for img in pipe.results()], pipe.scores.clone()))
        return out, pipe

    eager, pe = collect(False)
    graph, pg = collect(True)
    assert pe._graph is None and pg._graph is not None, 'the graph path did not capture'
    for it, ((da, sa), (db, sb)) in enumerate(zip(eager, graph)):
        assert torch.equal(sa, sb), it
        for ia, ib in zip(da, db):
            for a, b in zip(ia, ib):
                assert np.array_equal(a, b), it
    assert not torch.equal(eager[2][1], eager[5][1])            # the weight update changed the scores (same input)
