"""Optimizer / LR-schedule glue with the reference's entry points (utils/solver.py).

build_optimizer (:6-33): SGD with one parameter group per tensor; in phase 2 with method 'ours'
the groups are scaled by NAME — `base` in the key: lr x 0.1; `extras` or `Norm`: lr x 0.5 —
which is why the state-dict prefixes are part of the contract.  WarmupMultiStepLR (:49-111):
lr = base_lr * warmup(iter) * gamma ** (#milestones <= iter), linear warmup from `warmup_factor`.
"""
from bisect import bisect_right

import torch


def lr_multiplier(args, key):
    if args.phase == 2 and args.method == 'ours':
        if 'base' in key:
            return 0.1
        if 'extras' in key or 'Norm' in key:
            return 0.5
    return 1.0


def build_optimizer(args, model):
    groups = [{'params': [p], 'lr': args.lr * lr_multiplier(args, name), 'weight_decay': args.weight_decay}
              for name, p in model.named_parameters() if p.requires_grad]
    return torch.optim.SGD(groups, args.lr, momentum=args.momentum)


def _get_warmup_factor_at_iter(method, iter, warmup_iters, warmup_factor):
    if iter >= warmup_iters:
        return 1.0
    if method == 'constant':
        return warmup_factor
    if method == 'linear':
        alpha = iter / warmup_iters
        return warmup_factor * (1 - alpha) + alpha
    raise ValueError('Unknown warmup method: {}'.format(method))


class WarmupMultiStepLR(torch.optim.lr_scheduler._LRScheduler):
    def __init__(self, optimizer, milestones, gamma=0.1, warmup_factor=1e-6, warmup_iters=1000,
                 warmup_method='linear', last_epoch=-1):
        if list(milestones) != sorted(milestones):
            raise ValueError('Milestones should be a list of increasing integers. Got {}'.format(milestones))
        self.milestones, self.gamma = milestones, gamma
        self.warmup_factor, self.warmup_iters, self.warmup_method = warmup_factor, warmup_iters, warmup_method
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        w = _get_warmup_factor_at_iter(self.warmup_method, self.last_epoch, self.warmup_iters, self.warmup_factor)
        decay = self.gamma ** bisect_right(self.milestones, self.last_epoch)
        return [base * w * decay for base in self.base_lrs]

    def _compute_values(self):
        return self.get_lr()


def build_lr_scheduler(args, optimizer):
    return WarmupMultiStepLR(optimizer, args.steps, warmup_iters=args.warmup_iter)
