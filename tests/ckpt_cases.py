"""Checkpoint-loading scenarios shared by tools/gen_goldens.py (which runs them through the REFERENCE's
utils/checkpointer.py:169-207,259-297 and stores what happened in tests/golden/checkpointer.npz) and
tests/test_solver_checkpoint_cpu.py (which runs them through the build's utils/checkpointer.py and compares).

Only data construction and observation live here -- no checkpointer logic.  A scenario writes `.pth` files built
from the model's own state-dict keys with name-seeded values, calls the given classes, and records:
  changed        sorted model keys whose tensors differ after load()
  returned       sorted keys of the dict load() returned (+ the `iteration` value when present)
  opt_loaded     did the optimizer checkpointable receive the saved state
  n_ckpt         len(checkpointer.checkpointables) after load()
"""
import os
import types
import zlib

import torch


def _val(key, like, salt):
    g = torch.Generator().manual_seed(zlib.crc32((salt + key).encode()))
    if like.is_floating_point():
        return torch.randn(like.shape, generator=g)
    return torch.full(like.shape, 7, dtype=like.dtype)


def _snapshot(model):
    return {k: v.detach().clone() for k, v in model.state_dict().items()}


def _changed(model, before):
    return sorted(k for k, v in model.state_dict().items() if not torch.equal(v, before[k]))


def _opt(model):
    return torch.optim.SGD([p for p in model.parameters() if p.requires_grad], 0.1, momentum=0.9)


def _opt_state_with_momentum(model):
    opt = _opt(model)
    for p in opt.param_groups[0]['params'][:3]:
        p.grad = torch.ones_like(p)
    opt.step()
    return opt.state_dict()


def run(make_model, ck, tmp):
    """make_model(phase) -> RFBNet-300 (C=60, 'ours', 'transfer'); ck: module with DetectionCheckpointer /
    PeriodicCheckpointer; tmp: empty directory.  -> {name: list of strings} (stable, comparable)."""
    out = {}

    def args(phase, folder):
        return types.SimpleNamespace(phase=phase, save_folder=folder, method='ours', setting='transfer')

    def observe(tag, model, before, c, rest, opt):
        out[tag + '.changed'] = _changed(model, before)
        ret = sorted(rest.keys())
        out[tag + '.returned'] = ret + (['iteration=%d' % rest['iteration']] if 'iteration' in rest else [])
        out[tag + '.opt_loaded'] = [str(bool(opt is not None and len(opt.state_dict()['state']) > 0))]
        out[tag + '.n_ckpt'] = [str(len(c.checkpointables))]

    # A. phase 2 fine-tune start: a DataParallel-saved phase-1 file (every key `module.`-prefixed) whose conf heads
    #    have another class count (shape mismatch -> skipped), one key the model does not know, optimizer + iteration
    m = make_model(2)
    keys = [k for k in m.state_dict() if k.startswith(('base.0.', 'Norm.branch0.', 'extras.3.', 'conf.', 'loc.0.'))]
    sd = {'module.' + k: _val(k, m.state_dict()[k], 'A') for k in keys}
    for k in keys:
        if k.startswith('conf.') and k.endswith('weight'):
            v = m.state_dict()[k]
            sd['module.' + k] = torch.zeros((v.shape[0] // 3,) + tuple(v.shape[1:]))
    sd['module.not_in_model.weight'] = torch.zeros(3)
    path = os.path.join(tmp, 'phase1_dp.pth')
    torch.save({'model': sd, 'optimizer': _opt_state_with_momentum(m), 'iteration': 41, 'note': 'x'}, path)
    before, opt = _snapshot(m), _opt(m)
    c = ck.DetectionCheckpointer(m, args(2, tmp), optimizer=opt)
    observe('A', m, before, c, c.load(path), opt)

    # B. ImageNet trunk: a bare state dict (no 'model' wrapper) in a file named vgg16_reducedfc -> `base.` prefix
    m = make_model(1)
    base = {k[len('base.'):]: _val(k, v, 'B') for k, v in m.state_dict().items() if k.startswith('base.')}
    path = os.path.join(tmp, 'vgg16_reducedfc.pth')
    torch.save(base, path)
    before, opt = _snapshot(m), _opt(m)
    c = ck.DetectionCheckpointer(m, args(1, tmp), optimizer=opt)
    observe('B', m, before, c, c.load(path), opt)

    # C. phase 1 resume: optimizer state is restored, iteration comes back to the caller
    m = make_model(1)
    keys = [k for k in m.state_dict() if k.startswith(('extras.0.branch1.', 'obj.'))]
    path = os.path.join(tmp, 'resume.pth')
    torch.save({'model': {k: _val(k, m.state_dict()[k], 'C') for k in keys}, 'optimizer': _opt_state_with_momentum(m),
                'iteration': 7}, path)
    before, opt = _snapshot(m), _opt(m)
    c = ck.DetectionCheckpointer(m, args(1, tmp), optimizer=opt)
    observe('C', m, before, c, c.load(path), opt)

    # D. `module.` only on SOME keys: nothing is stripped, the prefixed keys are simply unknown
    m = make_model(1)
    sd = {'module.loc.1.bias': _val('loc.1.bias', m.state_dict()['loc.1.bias'], 'D'),
          'loc.2.bias': _val('loc.2.bias', m.state_dict()['loc.2.bias'], 'D')}
    path = os.path.join(tmp, 'mixed.pth')
    torch.save({'model': sd}, path)
    before = _snapshot(m)
    c = ck.DetectionCheckpointer(m, args(1, tmp))
    observe('D', m, before, c, c.load(path), None)

    # E. saving: file names of the periodic schedule, the tag file, the top-level layout of a checkpoint
    folder = os.path.join(tmp, 'save')
    os.makedirs(folder)
    small = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4))
    opt = _opt(small)
    c = ck.DetectionCheckpointer(small, args(1, folder), optimizer=opt)
    pc = ck.PeriodicCheckpointer(c, period=4, max_iter=10)
    for it in range(10):
        pc.step(it, lr=0.5)
    out['E.files'] = sorted(os.listdir(folder))
    out['E.tag'] = [open(os.path.join(folder, 'last_checkpoint')).read()]
    out['E.resume_path'] = [os.path.basename(c.get_checkpoint_file())]
    data = torch.load(os.path.join(folder, 'model_final.pth'), map_location='cpu')
    out['E.top_keys'] = sorted(data.keys())
    out['E.model_keys'] = sorted(data['model'].keys())
    out['E.extra'] = ['iteration=%d' % data['iteration'], 'lr=%g' % data['lr']]
    return out
