"""Data-parallel training path on ONE MI355X (BASELINE configs[3]): what `train.py:206-242,296-297`
does with DataParallel, here as one process per rank -- checked without an 8-GPU node:

  * the whole-network gradient against torch-CPU float64 autograd at 1e-4 (BatchNorm in eval mode, so the
    function is smooth and batch-size independent; the batch-statistics case is in test_gpu_train.py);
  * two processes on the same device (gloo): each runs the HIP backward on its half of a bs-8 batch through
    GradBucketer + the global loss normaliser; the averaged gradient equals the one-process full-batch one;
  * the engine's guards (stale BatchNorm folding, backward after a second forward, > 256 ground-truth boxes).
"""
import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import rel_err
from ctdet import _lib, ops, synth
from oracle import box_ref, loss_ref, rfbnet_ref

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module', autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail('the gpu tests need a HIP device; none visible')


def _net(size, C, phase=1, setting='transfer'):
    from models.RFB_Net_vgg import build_net
    net = build_net(types.SimpleNamespace(method='ours', phase=phase, setting=setting), size, C)
    net.load_state_dict(synth.fill_state_dict(net.state_dict()), strict=True)
    net = net.cuda()
    net.device = 'cuda'
    return net


def _freeze_bn(net):
    """What a fine-tuning script does to keep BatchNorm statistics fixed: the nn.BatchNorm2d modules in eval()."""
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
    return net


@pytest.mark.parametrize('form', ['f16x2', 'bf16x3'])
def test_frozen_bn_network_gradients_match_fp64_autograd(form, monkeypatch):
    """(Both operand forms of the Winograd launches: f16x2 is what a bs-8 step runs by default -- forward AND data gradients,
    maxima of |activation| / |dZ| from the producing kernels, train_engine._wire_absmax -- CTDET_TRAIN_H2=0 keeps bf16x3.)
    Every parameter gradient of RFBNet-300 (bs 8, BatchNorm in eval mode) against float64 autograd at 1e-4
    normalised.  The loss is a fixed random linear functional of (loc, conf, obj), so nothing but the network's own
    backward kernels (dgrad / wgrad direct + Winograd, bias/ReLU, frozen-BN, pools, head gather) is between the output
    gradient and the parameters.

    A ReLU network is only piecewise smooth: torch-CPU fp32 itself is 2e-3 .. 2e-2 away from float64 on these
    gradients (tools/grad_debug.py --frozen-bn --batch 8), because a handful of the ~1e8 pre-activations lie within
    fp32 rounding of zero and one flipped unit on a 19x19 .. 5x5 map moves its filter's gradient by a percent.  So
    the float64 evaluation differentiates the SAME linear piece the device evaluated: it replays the engine's plan
    (tests/emu_backend.py, itself checked against the oracle in test_surface_cpu.py and again below) with every
    ReLU replaced by the device's activation pattern and every max-pool taken at the device's arg-max.  The free comparison against the oracle is kept as a loose
    bound."""
    from emu_backend import replay_plan_autograd
    from ctdet.engine import H2_TILES
    monkeypatch.setenv('CTDET_TRAIN_H2', '1' if form == 'f16x2' else '0')
    B = 8
    net = _freeze_bn(_net(300, 20).train())
    x = synth.images(B, 300, 'randn', 2024)
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    out = net(x.cuda())
    g = torch.Generator().manual_seed(5)
    R = [torch.randn(t.shape, generator=g) / t.numel() ** 0.5 for t in out]
    loss = sum((t * r.cuda()).sum() for t, r in zip(out, R))
    loss.backward()
    trt = net.train_runtime(B)
    fwd_h2 = [n for n, s_ in trt.state.items() if s_.fwd.rt.get('wino') in H2_TILES]
    dgrad_h2 = [n for n, s_ in trt.state.items() if getattr(s_, 'dgrad_wino', None) is not None and s_.dgrad_tile in H2_TILES]
    if form == 'f16x2':
        assert trt.h2 and len(fwd_h2) >= 15 and len(dgrad_h2) >= 15, (fwd_h2, dgrad_h2)
        assert any(trt.state[n].dgrad_tile == 48 for n in dgrad_h2) and any(trt.state[n].dgrad_tile == 47 for n in dgrad_h2)
    else:
        assert not trt.h2 and not fwd_h2 and not dgrad_h2
    names = {id(p): n for n, p in net.named_parameters()}
    leaf = {i: sd[n].double().requires_grad_(True) for i, n in names.items()}

    def masks(st, off, cout):
        return (trt.bufs[st.dst][:, st.dst_coff + off:st.dst_coff + off + cout] > 0).cpu()
    got64 = replay_plan_autograd(trt.plan, leaf, x, masks, pool_inputs=lambda st: trt.bufs[st.src].cpu())
    for a, b, n in zip(out, got64, ('loc', 'conf', 'obj')):
        assert rel_err(a.detach().cpu().reshape(B, -1), b.detach().float()) < 1e-4, n
    sum((t * r.double().reshape(B, -1)).sum() for t, r in zip(got64, R)).backward()
    worst = {}
    for name, prm in net.named_parameters():
        assert prm.grad is not None, name
        e = rel_err(prm.grad.cpu().double(), leaf[id(prm)].grad)
        if e >= 1e-4:
            worst[name] = e
    assert not worst, ' '.join('%s:%.1e' % kv for kv in worst.items())
    # the oracle (free ReLUs, float64): same forward at 1e-4, gradients within the flip-limited band
    leaf_o = {k: v.double().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and 'running' not in k}
    sdo = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    sdo.update(leaf_o)
    oo = rfbnet_ref.forward(sdo, x.double(), 300, 20, raw=True)
    for a, b, n in zip(out, oo, ('loc', 'conf', 'obj')):
        assert rel_err(a.detach().cpu(), b.detach().float()) < 1e-4, n
    sum((t * r.double()).sum() for t, r in zip(oo, R)).backward()
    for name, prm in net.named_parameters():
        assert rel_err(prm.grad.cpu().double(), leaf_o[name].grad) < 5e-2, name
    # frozen statistics were not touched
    bn = net.Norm.branch0[0].bn
    assert int(bn.num_batches_tracked) == 0 and torch.equal(bn.running_mean.cpu(), sd['Norm.branch0.0.bn.running_mean'])


def test_batch_stat_bn_network_gradients_match_fp64_autograd():
    """the same check with BatchNorm in TRAIN mode (batch statistics, what train.py:222-229 runs):
    every parameter gradient of RFBNet-300 (bs 8) against float64 autograd over the replayed plan with the device's
    ReLU pattern and pool arg-max, BatchNorm differentiated through the batch mean and variance.  Held to 1e-4
    wherever a BatchNorm channel sees at least 200 samples; the layers on the 3x3 and 1x1 maps normalise over 72 and
    8 samples per channel, where the backward of the normalisation divides by a variance estimated from that few values
    and fp32 (device or torch-CPU) is 1e-3 .. 1e-1 from fp64 -- those (and the parameters whose gradient flows only
    through them) get the loose bound the free comparison uses."""
    from emu_backend import replay_plan_autograd
    B = 8
    net = _net(300, 20).train()
    x = synth.images(B, 300, 'randn', 2024)
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    out = net(x.cuda())
    g = torch.Generator().manual_seed(5)
    R = [torch.randn(t.shape, generator=g) / t.numel() ** 0.5 for t in out]
    sum((t * r.cuda()).sum() for t, r in zip(out, R)).backward()
    trt = net.train_runtime(B)
    names = {id(p): n for n, p in net.named_parameters()}
    leaf = {i: sd[n].double().requires_grad_(True) for i, n in names.items()}

    def masks(st, off, cout):
        return (trt.bufs[st.dst][:, st.dst_coff + off:st.dst_coff + off + cout] > 0).cpu()
    got64 = replay_plan_autograd(trt.plan, leaf, x, masks, pool_inputs=lambda st: trt.bufs[st.src].cpu(), batch_stats=True)
    for a, b, n in zip(out, got64, ('loc', 'conf', 'obj')):
        assert rel_err(a.detach().cpu().reshape(B, -1), b.detach().float()) < 1e-4, n
    sum((t * r.double().reshape(B, -1)).sum() for t, r in zip(got64, R)).backward()
    # parameters behind a BatchNorm that normalises over fewer than 200 samples (extras.4 .. extras.6: 5x5 .. 1x1 maps
    # at bs 8), and the head convolutions fed by those maps
    small = ('extras.3.', 'extras.4.', 'extras.5.', 'extras.6.', 'loc.4.', 'conf.4.', 'obj.4.', 'loc.5.', 'conf.5.', 'obj.5.',
             'extras.2.')
    worst, loose = {}, {}
    gmax = max(float(v.grad.abs().max()) for v in leaf.values())
    for name, prm in net.named_parameters():
        assert prm.grad is not None, name
        ref = leaf[id(prm)].grad
        if float(ref.abs().max()) < 1e-9 * gmax:
            # exactly cancelled by a following batch normalisation (the bias of a branch-final BasicConv feeds a conv +
            # BatchNorm whose mean subtraction removes any constant): the true gradient is 0, the device's is rounding
            assert float(prm.grad.abs().max()) < 1e-5 * gmax, (name, float(prm.grad.abs().max()), gmax)
            continue
        e = rel_err(prm.grad.cpu().double(), ref)
        if name.startswith(small):
            if e >= 5e-2:
                loose[name] = e
        elif e >= 1e-4:
            worst[name] = e
    assert not worst, ' '.join('%s:%.1e' % kv for kv in sorted(worst.items(), key=lambda kv: -kv[1])[:12])
    assert not loose, ' '.join('%s:%.1e' % kv for kv in loose.items())


# ------------------------------------------------------------------ two ranks on one device
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _priors():
    from layers.functions import PriorBox
    from data import VOC_300
    return PriorBox(VOC_300).forward()


def _dp_step(net, batch, x, targets, sync):
    from layers.modules.multibox_loss_combined import MultiBoxLoss_combined
    priors = _priors().cuda()
    crit = MultiBoxLoss_combined(21, 0.5, True, 0, True, 3, 0.5, False)
    crit.sync_normalizer = sync
    if sync:
        net.train_runtime(batch).enable_grad_sync(bucket_bytes=16 << 20)      # several buckets
    ld = crit(net(x.cuda()), priors, [t.cuda() for t in targets])
    sum(ld.values()).backward()
    torch.cuda.synchronize()
    flat = torch.cat([p.grad.flatten() for p in net.parameters() if p.grad is not None]).cpu()
    return flat, {k: float(v) for k, v in ld.items()}


def _dp_worker(rank, world, port, q):
    try:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        # one device per rank over RCCL as soon as the box has two; on a one-GPU box both ranks share it over gloo
        two = torch.cuda.device_count() >= world
        torch.cuda.set_device(rank if two else 0)
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl' if two else 'gloo', rank=rank, world_size=world)
        net = _freeze_bn(_net(300, 20).train())
        B = 8 // world
        x = synth.images(8, 300, 'randn', 31)[rank * B:(rank + 1) * B]
        tg = synth.targets(8, 21, 17)[rank * B:(rank + 1) * B]
        flat, ld = _dp_step(net, B, x, tg, True)
        nb = len(net.train_runtime(B).bucketer.bucket_span)
        props = torch.cuda.get_device_properties(torch.cuda.current_device())
        info = {'backend': dist.get_backend(), 'device': torch.cuda.current_device(),
                'pci_bus_id': getattr(props, 'pci_bus_id', None)}
        q.put((rank, flat.numpy(), ld, nb, None, info))
        dist.destroy_process_group()
    except Exception as e:      # surface the failure in the parent instead of a queue timeout
        import traceback
        q.put((rank, None, None, 0, traceback.format_exc(), None))
        raise


def test_two_rank_gradient_equals_full_batch_gradient():
    """train.py:296-297 (DataParallel splits ONE batch) + multibox_loss_combined.py:119-122 (N over the whole
    batch): two single-GPU ranks, each with half of the bs-8 batch, all-reduce (mean) of the flat gradient in
    buckets issued from inside the HIP backward, loss normaliser made global -- equals the bs-8 step of one
    process (to the tolerance a ReLU network allows between two batch sizes) and, to rounding, the same two
    half-batches accumulated by hand in one process.  On a box with two or more GPUs the ranks run on DISTINCT devices
    over RCCL (backend 'nccl'), which is then asserted; on the one-GPU test box they share the device over gloo."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    # single-process references, computed here while the ranks work:
    #  (a) the full bs-8 batch in one step (what DataParallel's scatter/gather computes);
    #  (b) the two halves as two bs-4 steps of ONE process, combined by hand with the global normaliser -- the same
    #      kernels on the same data as the ranks, so it must agree with the all-reduced result to rounding.
    X, TG = synth.images(8, 300, 'randn', 31), synth.targets(8, 21, 17)
    net = _freeze_bn(_net(300, 20).train())
    full, ld_full = _dp_step(net, 8, X, TG, False)
    halves, n_half = [], []
    for r in range(2):
        h = _freeze_bn(_net(300, 20).train())
        g, _ = _dp_step(h, 4, X[4 * r:4 * r + 4], TG[4 * r:4 * r + 4], False)
        _, conf_t, _ = ops.match_batched([t.cuda() for t in TG[4 * r:4 * r + 4]], _priors().cuda(), 0.5, [0.1, 0.2])
        n_half.append(float(((conf_t[:, :, 0] > 0).float() * conf_t[:, :, 1]).sum(1).long().sum()))
        halves.append(g.double() * n_half[-1])          # un-normalised gradient of the half
        del h
    by_hand = ((halves[0] + halves[1]) / (n_half[0] + n_half[1])).float()
    res = sorted([q.get(timeout=900) for _ in range(world)], key=lambda t: t[0])
    for p in ps:
        p.join(120)
    for r in res:
        assert r[4] is None, r[4]
    for p in ps:
        assert p.exitcode == 0
    if torch.cuda.device_count() >= world:           # the first multi-GPU box runs RCCL here without being asked
        assert [r[5]['backend'] for r in res] == ['nccl'] * world, res[0][5]
        assert len({r[5]['device'] for r in res}) == world
        ids = [r[5]['pci_bus_id'] for r in res]
        assert None in ids or len(set(ids)) == world, ids
    else:
        assert [r[5]['backend'] for r in res] == ['gloo'] * world
    g0, g1 = torch.from_numpy(res[0][1]), torch.from_numpy(res[1][1])
    assert res[0][3] >= 3                                    # really bucketed
    assert torch.equal(g0, g1)                               # both ranks hold the same averaged gradient
    assert g0.shape == full.shape == by_hand.shape
    gmax = float(full.abs().max())
    off, bad_hand, bad_full = 0, {}, {}
    for name, prm in net.named_parameters():
        if prm.grad is None:
            continue
        n = prm.numel()
        a, b, c = g0[off:off + n], by_hand[off:off + n], full[off:off + n]
        # (b): same kernels, same data.  Normalised per parameter, with a floor for the heads on the 3x3 / 1x1 maps
        # whose gradient is ~1e-6 of the others (no matched prior there in this batch)
        e = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-4 * gmax)
        if e >= 1e-5:
            bad_hand[name] = e
        # (a): another batch size runs other tile / split-K choices, its activations differ in the last bits, and a
        # ReLU or max-pool arg-max that flips moves a small-map gradient by up to a percent (see the float64 test above)
        e = float((a - c).abs().max()) / max(float(c.abs().max()), 1e-2 * gmax)
        if e >= 3e-2:
            bad_full[name] = e
        off += n
    assert off == full.numel()
    assert not bad_hand, ' '.join('%s:%.1e' % kv for kv in sorted(bad_hand.items(), key=lambda kv: -kv[1])[:10])
    assert not bad_full, ' '.join('%s:%.1e' % kv for kv in sorted(bad_full.items(), key=lambda kv: -kv[1])[:10])
    assert float((g0 - full).norm() / full.norm()) < 2e-3    # and as a whole vector
    # per-rank losses are normalised by N_global / world: their mean is the full-batch loss
    for k in ld_full:
        assert abs(0.5 * (res[0][2][k] + res[1][2][k]) - ld_full[k]) < 1e-4 * max(1.0, abs(ld_full[k])), k


# ------------------------------------------------------------------ guards
def test_backward_after_second_forward_raises():
    net = _net(300, 20).train()
    x = synth.images(2, 300, 'randn', 3).cuda()
    o1 = net(x)
    o2 = net(x * 0.5)
    with pytest.raises(_lib.CtdetError, match='overwrote'):
        (o1[0].sum() + o2[0].sum()).backward()
    o3 = net(x)                 # a fresh forward/backward pair still works
    o3[0].sum().backward()
    assert net.base[0].weight.grad is not None


def test_eval_runtime_refolds_batchnorm_after_a_training_forward():
    """A train-mode forward updates running_mean / running_var through raw pointers; the eval runtime built BEFORE
    must pick the new statistics up even when no parameter changed in between."""
    net = _net(300, 20).eval()
    x = synth.images(2, 300, 'randn', 8)
    with torch.no_grad():
        before = [t.clone() for t in net.forward_raw(x.cuda())]
    net.train()
    with torch.no_grad():
        for _ in range(3):
            net(x.cuda() * 3.0 + 1.0)
    net.eval()
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    with torch.no_grad():
        after = net.forward_raw(x.cuda())
        want = rfbnet_ref.forward(sd, x, 300, 20, raw=True)
    assert not torch.equal(before[1], after[1])
    for a, b, n in zip(after, want, ('loc', 'conf', 'obj')):
        assert rel_err(a.cpu(), b) < 1e-4, n


def test_match_more_than_256_ground_truth_boxes():
    """utils/box_utils.py:83-132 has no limit on the number of boxes per image."""
    priors = box_ref.prior_box(box_ref.ANCHOR_CFGS['VOC_300'])
    rng = np.random.RandomState(3)
    G = 700
    xy = rng.uniform(0, 0.8, (G, 2))
    wh = rng.uniform(0.05, 0.2, (G, 2))
    t = torch.from_numpy(np.concatenate([xy, xy + wh, rng.randint(1, 21, (G, 1)), np.ones((G, 1))], 1).astype(np.float32))
    batch = [t, t[:300], t[100:357], synth.targets(1, 21, 5)[0]]
    loc_t, conf_t, obj_t, ovl = ops.match_batched([b.cuda() for b in batch], priors.cuda(), 0.5, [0.1, 0.2], True)
    for i, b in enumerate(batch):
        wl, wc, wo, wov = box_ref.match(0.5, b[:, :4], priors, [0.1, 0.2], b[:, 4:])
        assert torch.equal(conf_t[i].cpu(), wc), i
        assert torch.equal(obj_t[i].cpu(), wo), i
        assert torch.equal(ovl[i].cpu(), wov), i
        np.testing.assert_allclose(loc_t[i].cpu().numpy(), wl.numpy(), rtol=1e-5, atol=2e-5)
