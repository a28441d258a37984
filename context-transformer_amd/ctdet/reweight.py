"""Initialisation of the OBJ(Target) classifier from support-set features: `init_reweight` of
train.py:252-286 on whole batches.

For every target class: all raw conf rows (model(x, init=True), [B,P,C]) of priors matched to that class over
the first `init_iter` batches are L2-normalised and averaged; the average, normalised again, becomes the class's
row of `OBJ_Target.weight` ('incre': only classes 16.. are new rows).  Matching is one `ct_match_batched` per
batch; the per-class sums are all-reduced when a process group is active (each rank sees its image shard).
"""
import torch
import torch.distributed as tdist

from . import ops


def class_feature_sums(conf_data, labels, num_classes):
    """conf_data [B,P,C] raw logits, labels [B,P] (0 = background) -> (sums [num_classes-1, C] of the
    row-normalised features per class, counts [num_classes-1])."""
    B, P, C = conf_data.shape
    lab = labels.reshape(-1).long()
    sel = lab > 0
    rows = conf_data.reshape(-1, C)[sel]
    rows = rows / rows.norm(dim=1, keepdim=True)
    idx = lab[sel] - 1
    sums = torch.zeros(num_classes - 1, C, device=conf_data.device, dtype=conf_data.dtype)
    counts = torch.zeros(num_classes - 1, device=conf_data.device, dtype=conf_data.dtype)
    sums.index_add_(0, idx, rows)
    counts.index_add_(0, idx, torch.ones_like(idx, dtype=conf_data.dtype))
    return sums, counts


def weights_from_sums(sums, counts, setting='transfer', num_base=15):
    """(train.py:280-286) mean of the normalised rows, normalised; classes without a sample give NaN rows
    exactly like the reference's mean over an empty tensor."""
    mean = sums / counts[:, None]
    if setting == 'incre':
        mean = mean[num_base:]
    return mean / mean.norm(dim=1, keepdim=True)


@torch.no_grad()
def init_reweight(model, priors, batches, num_classes, overlap_threshold=0.5, setting='transfer', init_iter=None,
                  variances=(0.1, 0.2)):
    """batches: iterable of (data [B,3,S,S], targets list of [G,6]).  Sets model.OBJ_Target.weight.data and
    returns it."""
    net = model.module if hasattr(model, 'module') else model
    dev = net._device()
    pri = priors.to(dev, torch.float32).contiguous()
    sums = counts = None
    for it, (data, targets) in enumerate(batches):
        if init_iter is not None and it >= init_iter:
            break
        conf = net(data.to(dev), init=True)
        _, conf_t, _ = ops.match_batched([t.to(dev) for t in targets], pri, overlap_threshold, variances)
        s, c = class_feature_sums(conf, conf_t[:, :, 0], num_classes)
        sums, counts = (s, c) if sums is None else (sums + s, counts + c)
    if sums is None:
        raise ValueError('init_reweight: no batches')
    if tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1:
        tdist.all_reduce(sums)
        tdist.all_reduce(counts)
    w = weights_from_sums(sums, counts, setting)
    net.OBJ_Target.weight.data = w.to(net.OBJ_Target.weight.dtype)
    return net.OBJ_Target.weight.data
