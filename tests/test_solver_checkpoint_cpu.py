"""solver LR groups / schedule against goldens produced by the reference's
utils/solver.py on the reference's RFBNet, and the checkpoint loading rules of
utils/checkpointer.py (module. strip, base. prefix, shape-mismatch skip, phase-2 weights-only,
resume tag), pinned by tests/golden/checkpointer.npz: the reference's DetectionCheckpointer /
PeriodicCheckpointer executed on the reference's RFBNet over the scenarios of tests/ckpt_cases.py
(tools/gen_goldens.py `checkpointer`), plus behaviour tests citing its lines."""
import os
import types

import numpy as np
import pytest
import torch

from models.RFB_Net_vgg import build_net
from utils import checkpointer as ck
from utils import solver

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'solver.npz'))
CASES = {'p2ours': ('ours', 2, 'transfer', 20), 'p1': ('ours', 1, 'transfer', 60), 'p2ft': ('ft', 2, 'incre', 20)}


def _args(method, phase, setting, **kw):
    return types.SimpleNamespace(method=method, phase=phase, setting=setting, lr=4e-3, weight_decay=5e-4,
                                 momentum=0.9, steps=[30, 50], warmup_iter=10, **kw)


@pytest.mark.parametrize('tag', sorted(CASES))
def test_optimizer_groups_and_schedule_match_reference(tag):
    method, phase, setting, C = CASES[tag]
    args = _args(method, phase, setting)
    net = build_net(args, 300, C)
    opt = solver.build_optimizer(args, net)
    names = [k for k, v in net.named_parameters() if v.requires_grad]
    assert names == [str(n) for n in G[tag + '_names']]            # group ORDER is what optimizer checkpoints bind to
    assert [g['params'][0].numel() for g in opt.param_groups] == G[tag + '_numel'].tolist()
    assert np.array_equal(np.array([g['lr'] for g in opt.param_groups]), G[tag + '_lr'])
    assert np.array_equal(np.array([g['weight_decay'] for g in opt.param_groups]), G[tag + '_wd'])
    assert opt.defaults['momentum'] == 0.9
    sched = solver.build_lr_scheduler(args, opt)
    rows = []
    for it in range(60):
        rows.append([opt.param_groups[0]['lr'], opt.param_groups[-1]['lr']])
        opt.step()
        sched.step()
    assert np.array_equal(np.array(rows), G[tag + '_sched'])


def test_phase2_ours_multipliers_by_prefix():
    args = _args('ours', 2, 'transfer')
    assert solver.lr_multiplier(args, 'base.0.weight') == 0.1
    assert solver.lr_multiplier(args, 'extras.1.branch0.0.conv.weight') == 0.5
    assert solver.lr_multiplier(args, 'Norm.ConvLinear.bn.bias') == 0.5
    assert solver.lr_multiplier(args, 'conf.0.weight') == 1.0
    assert solver.lr_multiplier(_args('ft', 2, 'transfer'), 'base.0.weight') == 1.0
    with pytest.raises(ValueError):
        solver.WarmupMultiStepLR(torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], 0.1), [5, 3])


def _small():
    m = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.Conv2d(4, 2, 1))
    return m


def test_save_resume_round_trip(tmp_path):
    args = _args('ours', 1, 'transfer', save_folder=str(tmp_path))
    m = _small()
    opt = torch.optim.SGD(m.parameters(), 0.1, momentum=0.9)
    c = ck.DetectionCheckpointer(m, args, optimizer=opt)
    assert not c.has_checkpoint() and c.resume_or_load('', resume=True) == {}
    pc = ck.PeriodicCheckpointer(c, period=5, max_iter=12)
    for it in range(12):
        pc.step(it)
    assert sorted(os.listdir(tmp_path)) == ['last_checkpoint', 'model_0000004.pth', 'model_0000009.pth', 'model_final.pth']
    assert c.get_checkpoint_file().endswith('model_final.pth')
    m2 = _small()
    c2 = ck.DetectionCheckpointer(m2, args, optimizer=torch.optim.SGD(m2.parameters(), 0.1, momentum=0.9))
    extra = c2.resume_or_load('ignored.pth', resume=True)
    assert extra == {'iteration': 11}
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    with pytest.raises(ValueError):
        c.save('sub/dir')


def test_load_rules(tmp_path):
    m = _small()
    src = {k: torch.randn_like(v) for k, v in m.state_dict().items()}
    # (a) DataParallel prefix on every key is stripped; wrong-shape key is skipped, the rest loads
    wrapped = {'module.' + k: v for k, v in src.items()}
    wrapped['module.1.weight'] = torch.zeros(7, 4, 1, 1)
    wrapped['module.extra'] = torch.zeros(1)
    torch.save({'model': wrapped, 'iteration': 3, 'optimizer': {'bogus': 1}}, tmp_path / 'a.pth')
    before = m.state_dict()['1.weight'].clone()
    opt = torch.optim.SGD(m.parameters(), 0.1)
    c = ck.DetectionCheckpointer(m, _args('ours', 2, 'transfer', save_folder=str(tmp_path)), optimizer=opt)
    rest = c.load(str(tmp_path / 'a.pth'))
    assert torch.equal(m.state_dict()['0.weight'], src['0.weight'])
    assert torch.equal(m.state_dict()['1.weight'], before)
    assert c.incompatible.missing_keys == ['1.weight'] and c.incompatible.unexpected_keys == ['extra']
    # phase 2: optimizer state untouched (left in the returned dict), iteration dropped
    assert 'iteration' not in rest and rest['optimizer'] == {'bogus': 1} and c.checkpointables == {}
    # (b) a prefix that is not on every key stays
    mixed = dict(src)
    mixed['module.foo'] = torch.zeros(1)
    torch.save(mixed, tmp_path / 'b.pth')                         # bare state dict, no 'model' wrapper
    c = ck.DetectionCheckpointer(m, _args('ours', 1, 'transfer', save_folder=str(tmp_path)))
    c.load(str(tmp_path / 'b.pth'))
    assert c.incompatible.unexpected_keys == ['module.foo'] and not c.incompatible.missing_keys


def test_vgg16_reducedfc_gets_base_prefix(tmp_path):
    args = _args('ours', 1, 'transfer', save_folder=str(tmp_path))
    net = build_net(args, 300, 60)
    base = {k[len('base.'):]: torch.randn_like(v) for k, v in net.state_dict().items() if k.startswith('base.')}
    assert len(base) == 30
    path = str(tmp_path / 'vgg16_reducedfc.pth')
    torch.save(base, path)
    c = ck.DetectionCheckpointer(net, args)
    c.load(path)
    assert all(torch.equal(net.state_dict()['base.' + k], v) for k, v in base.items())
    assert not c.incompatible.unexpected_keys and all(not k.startswith('base.') for k in c.incompatible.missing_keys)


def test_checkpointer_matches_the_reference_on_recorded_scenarios(tmp_path):
    """utils/checkpointer.py:169-207 (_load_model), :259-297 (DetectionCheckpointer.load), :48-71 (save), :300-349
    (PeriodicCheckpointer): same files, same model keys -> the same tensors change, the same dict comes back, the
    same checkpointables get loaded / dropped, the same files are written."""
    import ckpt_cases
    want = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'checkpointer.npz'))

    def make_model(phase):
        return build_net(types.SimpleNamespace(method='ours', phase=phase, setting='transfer'), 300, 60)
    got = ckpt_cases.run(make_model, ck, str(tmp_path))
    assert sorted(got) == sorted(want.files)
    for k in sorted(got):
        assert got[k] == [str(v) for v in want[k]], k
    assert len(got['A.changed']) == 28 and got['A.n_ckpt'] == ['0'] and got['C.opt_loaded'] == ['True']
