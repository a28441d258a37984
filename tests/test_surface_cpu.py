"""CPU-side checks of the drop-in surface: module tree / state-dict contract, host-side
functions, the engine plan wiring (replayed with tests/emu_backend.py against the oracle), and
the C ABI (symbols + the host-only entry points).  No GPU needed."""
import ctypes
import os
import re
import types

import numpy as np
import pytest
import torch

from conftest import REPO, rel_err
from ctdet import _lib, engine, ops, synth
from oracle import box_ref, nms_ref, rfbnet_ref

torch.set_num_threads(8)


def _args(phase=1, setting='transfer', method='ours'):
    return types.SimpleNamespace(method=method, phase=phase, setting=setting)


def _net(size, C, phase=1, setting='transfer'):
    from models.RFB_Net_vgg import build_net
    net = build_net(_args(phase, setting), size, C)
    net.load_state_dict(synth.fill_state_dict(net.state_dict()), strict=True)
    return net.eval()


# ---------------------------------------------------------------- C ABI
def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(REPO, 'include', 'ctdet.h')).read()
    declared = set(re.findall(r'\b(ct_[a-z0-9_]+)\s*\(', hdr))
    declared -= {'ct_stream_t'}
    assert len(declared) >= 25
    handle = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(handle, s)]
    assert not missing, missing
    assert set(_lib.SIGNATURES) == declared
    assert _lib.lib().ct_abi_version() == 1
    assert _lib.lib().ct_conv_num_configs() >= 4
    assert _lib.lib().ct_conv_kpad(512, 3, 3) % 36 == 0
    assert _lib.lib().ct_conv_kpad(3, 3, 3) == 36


def test_device_ops_refuse_cpu_tensors():
    with pytest.raises(_lib.CtdetError):
        ops.decode(torch.zeros(4, 4), torch.ones(4, 4), [0.1, 0.2])
    net = _net(300, 20)
    net.device = 'cpu'
    with pytest.raises(_lib.CtdetError):
        net(torch.zeros(1, 3, 300, 300))


def test_host_cpu_nms_matches_oracle(golden):
    g = golden('nms.npz')
    from utils.nms.cpu_nms import cpu_nms, cpu_soft_nms
    from utils.nms.py_cpu_nms import py_cpu_nms
    for ci in range(int(g['ncases'])):
        d = g['c%d_dets' % ci]
        for thr in (0.45, 0.3, 0.5, 0.7):
            tag = 'c%d_t%02d' % (ci, int(round(thr * 100)))
            assert list(ops.cpu_nms(d, thr, ge=False)) == list(g[tag + '_gt']), tag
            assert py_cpu_nms(d, thr) == list(g[tag + '_gt']), tag
            assert cpu_nms(d, thr) == list(nms_ref.nms(d, thr, ge=True)), tag
            if bool(g['have_ge']):
                assert cpu_nms(d, thr) == list(g[tag + '_ge']), tag
    assert cpu_nms(g['eq_dets'], 0.5) == [0]
    assert list(ops.cpu_nms(g['eq_dets'], 0.5, ge=False)) == [0, 1]
    if bool(g['have_ge']):
        for m in (0, 1, 2):
            b = g['c6_dets'].copy()
            keep = cpu_soft_nms(b, 0.5, 0.3, 0.001, m)
            assert len(keep) == int(g['soft_m%d_n' % m])
            n = len(keep)
            assert np.array_equal(b[:n, :4], g['soft_m%d_boxes' % m][:n, :4])
            np.testing.assert_allclose(b[:n, 4], g['soft_m%d_boxes' % m][:n, 4], rtol=2e-6)
    from utils.nms_wrapper import nms
    assert nms(np.zeros((0, 5), np.float32), 0.45) == []


# ---------------------------------------------------------------- surface
def test_prior_box_and_config(golden):
    import hashlib
    from data import config as cfg
    from layers.functions import PriorBox
    g = golden('box_ops.npz')
    for name in box_ref.ANCHOR_CFGS:
        assert getattr(cfg, name) == box_ref.ANCHOR_CFGS[name], name
        p = PriorBox(getattr(cfg, name)).forward().numpy()
        sha = np.frombuffer(hashlib.sha256(p.tobytes()).digest(), dtype=np.uint8)
        assert (sha == g['prior_%s__sha' % name]).all(), name


@pytest.mark.parametrize('fname,size,C,phase,setting', [
    ('rfb300_phase1.npz', 300, 20, 1, 'transfer'),
    ('rfb300_phase2_transfer.npz', 300, 60, 2, 'transfer'),
    ('rfb300_phase2_incre.npz', 300, 15, 2, 'incre'),
    ('rfb512_phase1.npz', 512, 20, 1, 'transfer')])
def test_state_dict_keys_match_reference(golden, fname, size, C, phase, setting):
    from models.RFB_Net_vgg import build_net
    g = golden(fname)
    net = build_net(_args(phase, setting), size, C)
    sd = net.state_dict()
    assert sorted(sd.keys()) == sorted(g['keys'].tolist())
    shapes = rfbnet_ref.param_shapes(size, C, phase, 'ours', setting)
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    assert sum(p.numel() for p in net.parameters()) == int(g['nparams'])
    names = [n for n, _ in net.named_parameters()]
    assert any(n.startswith('base.') for n in names) and any(n.startswith('Norm.') for n in names)
    if phase == 2:
        assert (net.Wz == 0).all() and float(net.scale) == 5.0 and not net.scale.requires_grad


# ---------------------------------------------------------------- plan wiring
@pytest.mark.parametrize('size,C,phase,setting,B', [(300, 20, 1, 'transfer', 1), (300, 15, 2, 'incre', 1),
                                                     (512, 20, 1, 'transfer', 1)])
def test_plan_wiring_against_oracle(size, C, phase, setting, B):
    from emu_backend import EmuBackend
    net = _net(size, C, phase, setting)
    rt = engine.Runtime(net, B, EmuBackend(), tune=False)
    x = synth.images(B, size, 'randn', 1234)
    sd = {k: v for k, v in net.state_dict().items()}
    with torch.no_grad():
        loc, conf, obj = rt.run_backbone(x)
        want = rfbnet_ref.forward(sd, x, size, C, 1, 'ft', raw=True)     # raw heads, no ctx block
        assert not torch.isnan(loc).any() and not torch.isnan(conf).any() and not torch.isnan(obj).any()
        assert rel_err(loc.view(B, -1, 4), want[0]) < 2e-5
        assert rel_err(conf.view(B, -1, C), want[1]) < 2e-5
        assert rel_err(obj.view(B, -1, 2), want[2]) < 2e-5
        if phase == 2:
            pool = rt.bufs['pool']
            cpool = []
            srcs = rfbnet_ref.backbone(sd, x, size)
            for i, s in enumerate(srcs):
                c = torch.nn.functional.conv2d(s, sd['conf.%d.weight' % i], sd['conf.%d.bias' % i], 1, 1)
                k = rfbnet_ref.CTX_POOL[size][i]
                cpool.append(torch.nn.functional.max_pool2d(c, k, k, ceil_mode=True).permute(0, 2, 3, 1).reshape(B, -1))
            cpool = torch.cat(cpool, 1)
            assert pool.shape == cpool.shape
            assert rel_err(pool, cpool) < 2e-5
    plan = rt.plan
    if size == 300:
        assert plan.P == 11620
        assert abs(plan.conv_flops() / B - 72.24e9) < 0.3e9 or C != 20      # SURVEY 8d figure
        if phase == 2:
            assert plan.M == 1858 // 1 if C else True
    else:
        assert plan.P == 32756


@pytest.mark.parametrize('tag,size,C,phase,setting', [('300_p2_transfer', 300, 60, 2, 'transfer'),
                                                       ('300_p2_incre', 300, 15, 2, 'incre'), ('512_p1', 512, 20, 1, 'transfer')])
def test_init_weight_rules_match_reference(golden, tag, size, C, phase, setting):
    """models/RFB_Net_vgg.py:297-314 + :157-188 through the reference's own constructor (tests/golden/init.npz):
    every state-dict entry is initialised by the same RULE -- all zeros (every bias, Wz, fc_base), all ones (BN
    gamma, running_var), kaiming-normal fan_out (BasicConv convs, theta/phi/g), torch's default uniform (the plain
    nn.Conv2d of base / heads, OBJ_Target) -- with the same spread, and the same entries are trainable."""
    from models.RFB_Net_vgg import build_net
    g = golden('init.npz')
    torch.manual_seed(5)
    net = build_net(types.SimpleNamespace(method='ours', phase=phase, setting=setting), size, C)
    keys = [str(k) for k in g[tag + '_keys']]
    sd = net.state_dict()
    assert list(sd.keys()) == keys
    trainable = {k: v.requires_grad for k, v in net.named_parameters()}
    for k, (std, mean, amax, numel, rg) in zip(keys, g[tag + '_stats']):
        a = sd[k].double()
        assert a.numel() == int(numel), k
        assert bool(rg) == bool(trainable.get(k, False)), k
        if amax == 0.0:                                      # zeros rule
            assert float(a.abs().max()) == 0.0, k
        elif std == 0.0 and a.numel() > 1:                   # constant rule (ones) / scalar entries
            assert float(a.std()) == 0.0 and float(a.mean()) == mean, k
        elif a.numel() == 1:
            assert float(a.mean()) == mean, k                # `scale` = 5
        else:
            n = a.numel()
            tol = 6.0 / n ** 0.5 + 1e-3                     # sampling spread of a std estimate
            assert abs(float(a.std()) / std - 1.0) < tol, (k, float(a.std()), std)
            assert abs(float(a.mean())) < 6.0 * std / n ** 0.5 + 1e-12, k
            # max|x| / std: sqrt(3) for a uniform law, > 3.5 for 4096+ normal samples
            uniform_ref, uniform_own = amax / std < 2.0, float(a.abs().max()) / float(a.std()) < 2.0
            if n >= 4096:                                    # bounded (uniform) vs unbounded (normal) family
                assert uniform_ref == uniform_own, (k, amax / std, float(a.abs().max()) / float(a.std()))
    if phase == 2:
        gen = torch.Generator().manual_seed(77)
        net.OBJ_Target.weight.data = torch.randn(net.OBJ_Target.weight.shape, generator=gen)
        net.normalize()
        assert np.array_equal(net.OBJ_Target.weight.data.numpy(), g[tag + '_normalized'])     # :316-318, same expression
        assert torch.allclose(net.OBJ_Target.weight.data.norm(dim=1), torch.ones(net.OBJ_Target.weight.shape[0]), atol=1e-6)


def test_committed_tune_table_is_consistent():
    """ctdet/conv_tune_gfx950.json: every value names a kernel choice the engine knows; a dilated layer that the table gives
    to the three-kernel Winograd form keeps its previous choice under '|alt' (what a runtime with an accuracy policy, or one
    without tile 44, falls back to -- engine.apply_tuned), and no fallback is itself a Winograd name."""
    import json
    import re
    from ctdet import engine
    table = json.load(open(engine.TUNE_TABLE))
    wino = set(engine.WINO_NAME.values())
    for key, val in table.items():
        assert isinstance(val, str) and val, key
        if key.endswith('|alt') or key.endswith('|f32'):
            assert val not in wino, (key, val)
            assert key.rsplit('|', 1)[0] in table, key
            continue
        if key.endswith('|h2'):
            # the choice of a runtime on the f16x2 operand forms where it differs from the bf16x3 one (tools/tune_convs.py --h2):
            # an F(4x4,3x3) family by its bf16x3 name (mapped to the f16x2 twin at plan time), or an f16x2 tile of the direct kernel
            base = key.rsplit('|', 1)[0]
            assert base in table and (val in ('wino4s', 'wino4f') or re.fullmatch(r'h2:\d+x\d+k\d+d', val)), (key, val)
            if val.startswith('h2:'):
                assert table[base] == 'x3:' + val[3:], (key, val, table[base])
            elif int(re.search(r'_d(\d+)_', base).group(1)) > 1:
                assert val == 'wino4s', (key, val)
            continue
        m = re.match(r'(\d+)x(\d+)_s(\d+)_d(\d+)_c(\d+)_m(\d+)_(\d+)x(\d+)_b(\d+)', key)
        assert m, key
        kh, kw, stride, dil, cin = (int(m.group(i)) for i in (1, 2, 3, 4, 5))
        if val in wino:
            assert (kh, kw, stride) == (3, 3, 1), (key, val)
            if dil > 1:
                assert val == 'wino4s' and cin % 16 == 0 and key + '|alt' in table, (key, val)


def test_operand_form_policy_host_logic(monkeypatch):
    """engine.operand_form_h2 / ctx_policy (which operand form a runtime's Winograd and direct layers run; host logic, no device):
    plain networks switch to f16x2 from batch x size^2 >= 8 x 300^2 up, networks with the Context-Transformer block follow their
    policy ('h2': Winograd forms on f16x2 at every batch size, direct layers never; a tile list: neither), CTDET_H2 overrides."""
    for k in ('CTDET_H2', 'CTDET_H2_X3', 'CTDET_CTX_TILES'):
        monkeypatch.delenv(k, raising=False)
    plain300 = types.SimpleNamespace(method='ours', phase=1, size=300)
    plain512 = types.SimpleNamespace(method='ours', phase=1, size=512)
    ctx300 = types.SimpleNamespace(method='ours', phase=2, size=300)
    f = engine.operand_form_h2
    assert engine.ctx_policy(plain300) is None and engine.ctx_policy(ctx300) == 'h2'
    assert f(plain300, 4) == (False, False) and f(plain300, 8) == (True, True) and f(plain300, 32) == (True, True)
    assert f(plain512, 2) == (False, False) and f(plain512, 4) == (True, True)          # 4 x 512^2 > 8 x 300^2
    assert f(ctx300, 2) == (True, False) and f(ctx300, 32) == (True, False)
    monkeypatch.setenv('CTDET_H2_X3', '0')
    assert f(plain300, 32) == (True, False)
    monkeypatch.delenv('CTDET_H2_X3')
    monkeypatch.setenv('CTDET_H2', '0')
    assert f(plain300, 32) == (False, False) and f(ctx300, 32) == (False, False)
    monkeypatch.setenv('CTDET_H2', '2')
    assert f(plain300, 1) == (True, True)
    monkeypatch.delenv('CTDET_H2')
    monkeypatch.setenv('CTDET_CTX_TILES', '2,23')
    assert f(ctx300, 32) == (False, False) and engine.ctx_tile_set(ctx300) is not None
    monkeypatch.setenv('CTDET_CTX_TILES', 'any')
    assert f(ctx300, 32) == (True, True) and f(ctx300, 2) == (False, False) and engine.ctx_tile_set(ctx300) is None
    # the table maps its bf16x3 names onto the f16x2 twins, never the other way round
    assert engine.H2_OF_TILE == {44: 47, 46: 48} and set(engine.H2_TILES) == {47, 48}
    assert set(engine.WINO_NAME) == {2, 4, 23, 44, 46, 47, 48}                         # codes 24 / 45 are gone
