"""ORACLE (test infrastructure only): `init_reweight` of train.py:252-286, restated per image and per class
exactly in the reference's order (python loop over images with box_ref.match, boolean-mask gathers, cat)."""
import torch

from . import box_ref


def init_reweight(conf_batches, target_batches, priors, num_classes, overlap_threshold=0.5, setting='transfer'):
    """conf_batches: list of [B,P,C] raw conf (model(data, init=True)); target_batches: list of lists of [G,6].
    -> OBJ_Target.weight [T, C]."""
    cls_list = [torch.empty(0) for _ in range(num_classes - 1)]
    for conf_data, targets in zip(conf_batches, target_batches):
        num = conf_data.shape[0]
        conf_t = torch.zeros(num, priors.shape[0], 2)
        for idx in range(num):
            truths, labels = targets[idx][:, :-2], targets[idx][:, -2:]
            conf_t[idx] = box_ref.match(overlap_threshold, truths, priors, [0.1, 0.2], labels)[1]      # :272-276
        lists = [conf_data[conf_t[:, :, 0] == i] for i in range(1, num_classes)]                       # :278
        cls_list = [torch.cat((cls_list[i], lists[i]), 0) for i in range(num_classes - 1)]             # :279
    cls_list = [(item / item.norm(dim=1, keepdim=True)).mean(0) for item in cls_list]                   # :280
    if setting == 'incre':
        cls_list = cls_list[15:]                                                                         # :281-282
    return torch.stack([item / item.norm() for item in cls_list], 0)                                    # :284-286
