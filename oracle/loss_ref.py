"""Oracle (test infrastructure, NOT product): MultiBoxLoss_combined on torch-CPU fp32.

Restates layers/modules/multibox_loss_combined.py:42-124 (targets [G,6] =
[x1,y1,x2,y2,label,weight]; variance hard-coded [0.1,0.2] at :40) using
oracle.box_ref.match.  Differentiable w.r.t. (loc, conf, obj).
"""
import torch
import torch.nn.functional as F

from . import box_ref


def multibox_loss_combined(predictions, priors, targets, num_classes, threshold=0.5, negpos_ratio=3):
    loc_data, conf_data, obj_data = predictions
    num, num_priors = loc_data.shape[0], priors.shape[0]
    loc_t = torch.zeros(num, num_priors, 4)
    conf_t = torch.zeros(num, num_priors, 2)
    obj_t = torch.zeros(num, num_priors, dtype=torch.bool)
    for idx in range(num):                                           # :70-74
        t = targets[idx]
        l, c, o, _ = box_ref.match(threshold, t[:, :-2], priors, [0.1, 0.2], t[:, -2:])
        loc_t[idx], conf_t[idx], obj_t[idx] = l, c, o
    pos = conf_t[:, :, 0] > 0                                        # :76
    num_pos = (conf_t[:, :, 1] * pos.float()).sum(1, keepdim=True).long()
    loss_l = F.smooth_l1_loss(loc_data[pos], loc_t[pos], reduction='none')   # :81-85
    weight_pos = conf_t[pos][:, 1]
    loss_l = torch.sum(torch.sum(loss_l, dim=1) * weight_pos)
    with torch.no_grad():                                            # :88-96
        lo = F.cross_entropy(obj_data.reshape(-1, 2), obj_t.long().view(-1), reduction='none')
        lo[obj_t.view(-1)] = 0
        lo = lo.view(num, -1)
        _, loss_idx = lo.sort(1, descending=True)
        _, idx_rank = loss_idx.sort(1)
        num_neg = torch.clamp(negpos_ratio * num_pos, max=num_priors - 1)
        neg = idx_rank < num_neg.expand_as(idx_rank)
    mask = pos | neg                                                 # :99-101
    weight = conf_t[mask][:, 1]
    loss_obj = torch.sum(F.cross_entropy(obj_data[mask], obj_t[mask].long(), reduction='none') * weight)
    batch_conf = conf_data.reshape(-1, num_classes - 1)              # :106-117
    batch_obj = obj_data.reshape(-1, 2)
    logit_0 = batch_obj[:, 0].unsqueeze(1) + torch.log(torch.exp(batch_conf).sum(dim=1, keepdim=True))
    logit_k = batch_obj[:, 1].unsqueeze(1).expand_as(batch_conf) + batch_conf
    logit = torch.cat((logit_0, logit_k), 1).view(num, -1, num_classes)
    loss_c = torch.sum(F.cross_entropy(logit[mask], conf_t[mask][:, 0].long(), reduction='none') * weight)
    N = num_pos.sum()                                                # :119-122
    return {'loss_box_reg': loss_l / N, 'loss_cls': loss_c / N, 'loss_obj': loss_obj / N}
