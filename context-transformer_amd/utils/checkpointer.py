"""Checkpoint I/O with the reference's entry points (utils/checkpointer.py): `Checkpointer`,
`DetectionCheckpointer(model, args, **checkpointables)`, `PeriodicCheckpointer`.

File format: torch.save({'model': state_dict, <checkpointable>: state_dict..., **extra}) as
`<save_dir>/<name>.pth` plus a `last_checkpoint` tag file (utils/checkpointer.py:48-71,145-154).
Loading rules that matter for this build's RFBNet (same state-dict keys as the reference):
  * a leading `module.` on every key is stripped (:180, :387-417);
  * keys whose shape differs from the model's are skipped with a warning (:183-195);
  * load_state_dict(strict=False), missing/unexpected keys are logged (:197-207);
  * a path containing `vgg16_reducedfc` is a bare VGG state dict -> keys get `base.` (:282-283);
  * phase 2 loads weights only: optimizer/scheduler/iteration are ignored (:285-290).
The engine re-packs weights lazily (tensor version counters), so nothing else is needed after a load.
"""
import logging
import os
import pickle
from collections import OrderedDict

import torch
from torch.nn.parallel import DataParallel, DistributedDataParallel

TAG = 'last_checkpoint'


def _strip_prefix_if_present(state_dict, prefix):
    """Remove `prefix` only when EVERY key carries it (utils/checkpointer.py:387-417)."""
    keys = list(state_dict.keys())
    if not keys or not all(k.startswith(prefix) for k in keys):
        return
    for k in keys:
        state_dict[k[len(prefix):]] = state_dict.pop(k)
    meta = getattr(state_dict, '_metadata', None)
    if meta is not None:
        for k in list(meta.keys()):
            if len(k) == 0:
                continue
            meta[k[len(prefix):]] = meta.pop(k)


class Checkpointer(object):
    def __init__(self, model, save_dir='', *, save_to_disk=True, **checkpointables):
        if isinstance(model, (DistributedDataParallel, DataParallel)):
            model = model.module
        self.model = model
        self.checkpointables = dict(checkpointables)
        self.logger = logging.getLogger('Context-Transformer.' + __name__)
        self.save_dir, self.save_to_disk = save_dir, save_to_disk

    # ---- saving
    def save(self, name, **kwargs):
        if not self.save_dir or not self.save_to_disk:
            return
        data = {'model': self.model.state_dict()}
        for key, obj in self.checkpointables.items():
            data[key] = obj.state_dict()
        data.update(kwargs)
        basename = '{}.pth'.format(name)
        path = os.path.join(self.save_dir, basename)
        if os.path.basename(path) != basename:
            raise ValueError('checkpoint name must not contain a directory: %r' % name)
        self.logger.info('Saving checkpoint to {}'.format(path))
        with open(path, 'wb') as f:
            torch.save(data, f)
        self.tag_last_checkpoint(basename)

    def tag_last_checkpoint(self, basename):
        with open(os.path.join(self.save_dir, TAG), 'w') as f:
            f.write(basename)

    def has_checkpoint(self):
        return os.path.exists(os.path.join(self.save_dir, TAG))

    def get_checkpoint_file(self):
        try:
            with open(os.path.join(self.save_dir, TAG), 'r') as f:
                return os.path.join(self.save_dir, f.read().strip())
        except IOError:
            return ''

    # ---- loading
    def resume_or_load(self, path, *, resume=True):
        if resume and self.has_checkpoint():
            path = self.get_checkpoint_file()
        return self.load(path)

    def load(self, path):
        if not path:
            self.logger.info('No checkpoint found. Initializing model from scratch')
            return {}
        self.logger.info('Loading checkpoint from {}'.format(path))
        checkpoint = self._load_file(path)
        self._load_model(checkpoint)
        self._load_checkpointables(checkpoint, path)
        return checkpoint

    def _load_checkpointables(self, checkpoint, path):
        for key, obj in self.checkpointables.items():
            if key in checkpoint:
                self.logger.info('Loading {} from {}'.format(key, path))
                obj.load_state_dict(checkpoint.pop(key))

    def _load_file(self, f):
        return torch.load(f, map_location=torch.device('cpu'))

    def _load_model(self, checkpoint):
        state = checkpoint.pop('model')
        _strip_prefix_if_present(state, 'module.')
        own = self.model.state_dict()
        for k in list(state.keys()):
            if k in own and tuple(own[k].shape) != tuple(state[k].shape):
                self.logger.warning("'{}' has shape {} in the checkpoint but {} in the model! Skipped.".format(
                    k, tuple(state[k].shape), tuple(own[k].shape)))
                state.pop(k)
        for k, v in list(state.items()):
            if not isinstance(v, torch.Tensor):        # numpy arrays from .pkl model zoos
                state[k] = torch.as_tensor(v)
        incompatible = self.model.load_state_dict(state, strict=False)
        if incompatible.missing_keys:
            self.logger.info('Keys of the model not found in the checkpoint: %s', ', '.join(incompatible.missing_keys))
        if incompatible.unexpected_keys:
            self.logger.info('Keys of the checkpoint not used by the model: %s',
                             ', '.join(incompatible.unexpected_keys))
        self.incompatible = incompatible


class DetectionCheckpointer(Checkpointer):
    def __init__(self, model, args, *, save_to_disk=True, **checkpointables):
        super().__init__(model, args.save_folder, save_to_disk=save_to_disk, **checkpointables)
        self.phase = args.phase

    def _load_file(self, filename):
        if filename.endswith('.pkl'):                  # utils/checkpointer.py:224-238 model-zoo pickles
            with open(filename, 'rb') as f:
                data = pickle.load(f, encoding='latin1')
            if 'model' in data and '__author__' in data:
                return data
            if 'blobs' in data:
                data = data['blobs']
            data = {k: v for k, v in data.items() if not k.endswith('_momentum')}
            return {'model': data, '__author__': 'Caffe2', 'matching_heuristics': True}
        loaded = super()._load_file(filename)
        return loaded if 'model' in loaded else {'model': loaded}

    def load(self, path):
        if not path:
            self.logger.info('No checkpoint found. Initializing model from scratch')
            return {}
        self.logger.info('Loading checkpoint from {}'.format(path))
        checkpoint = self._load_file(path)
        if 'vgg16_reducedfc' in path:
            checkpoint['model'] = OrderedDict(('base.' + k, v) for k, v in checkpoint['model'].items())
        self._load_model(checkpoint)
        if self.phase == 2:                            # fine-tuning starts a fresh schedule
            self.checkpointables = {}
            checkpoint.pop('iteration', None)
        self._load_checkpointables(checkpoint, path)
        return checkpoint


class PeriodicCheckpointer(object):
    """utils/checkpointer.py:300-349: `model_{iteration:07d}` every `period`, `model_final` at the end."""

    def __init__(self, checkpointer, period, max_iter=None):
        self.checkpointer, self.period, self.max_iter = checkpointer, int(period), max_iter

    def step(self, iteration, **kwargs):
        iteration = int(iteration)
        state = dict(iteration=iteration, **kwargs)
        if (iteration + 1) % self.period == 0:
            self.checkpointer.save('model_{:07d}'.format(iteration), **state)
        if iteration >= self.max_iter - 1:
            self.checkpointer.save('model_final', **state)

    def save(self, name, **kwargs):
        self.checkpointer.save(name, **kwargs)
