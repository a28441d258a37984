"""MultiBoxLoss_combined -- drop-in for layers/modules/multibox_loss_combined.py:42-124.

Target assignment (jaccard / match / encode for the whole batch) is ONE call into the HIP
library (`ct_match_batched`) instead of a Python loop over images and ground truths; the loss
arithmetic itself (smooth-L1, two cross-entropies, 3:1 hard-negative ranking) is expressed on
the device tensors with autograd so it stays differentiable w.r.t. the predictions.
targets: list of [G,6] tensors = [x1,y1,x2,y2,label,mixup_weight].
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from collections import namedtuple

from ctdet import ops

# what match() assigns to every prior: encoded box [B,P,4], (label, mixup weight) [B,P,2], ignore mask [B,P] bool
MatchedTargets = namedtuple('MatchedTargets', 'loc_t conf_t obj_t')


class MultiBoxLoss_combined(nn.Module):
    def __init__(self, num_classes, overlap_thresh, prior_for_matching, bkg_label, neg_mining, neg_pos,
                 neg_overlap, encode_target):
        super().__init__()
        self.num_classes = num_classes
        self.threshold = overlap_thresh
        self.background_label = bkg_label
        self.encode_target = encode_target
        self.use_prior_for_matching = prior_for_matching
        self.do_neg_mining = neg_mining
        self.negpos_ratio = neg_pos
        self.neg_overlap = neg_overlap
        self.variance = [0.1, 0.2]
        self.sync_normalizer = False      # set True under multi-process data parallelism

    @torch.no_grad()
    def match(self, priors, targets, device=None, out=None):
        """The target assignment of :60-76 alone (it depends on the ground truth and the priors only): -> MatchedTargets,
        which forward() accepts in place of the raw target list.  `out`: a MatchedTargets of the same batch to overwrite
        in place -- a training step captured as a hipGraph (bench.py --train) reads the same three tensors at every
        replay, the matching itself (a host-built offset table, an H2D copy) stays outside the graph."""
        dev = torch.device(device) if device is not None else priors.device
        loc_t, conf_t, obj_t = ops.match_batched([t.to(dev) for t in targets], priors.to(dev).float().contiguous(),
                                                 self.threshold, self.variance)
        if out is None:
            return MatchedTargets(loc_t, conf_t, obj_t)
        out.loc_t.copy_(loc_t); out.conf_t.copy_(conf_t); out.obj_t.copy_(obj_t)
        return out

    def forward(self, predictions, priors, targets):
        loc_data, conf_data, obj_data = predictions
        dev = loc_data.device
        num, num_priors = loc_data.size(0), priors.size(0)
        if isinstance(targets, MatchedTargets):
            loc_t, conf_t, obj_t = targets
        else:
            loc_t, conf_t, obj_t = self.match(priors, targets, dev)
        labels, weights = conf_t[:, :, 0], conf_t[:, :, 1]
        pos = labels > 0
        num_pos = (weights * pos.float()).sum(1, keepdim=True).long()
        # Every term below is a MASKED SUM over all priors instead of the reference's boolean gather (`x[pos]`,
        # `x[pos | neg]`): the same numbers (a masked-out row contributes exactly 0 to the value and to the gradient),
        # but no `nonzero` behind the indexing, i.e. no host synchronisation in the middle of the training step -- the
        # host keeps issuing the backward pass while the forward still runs (tools/train_bench.py at small batches).

        # localisation: smooth-L1 on positives, weighted by the mixup weight (:81-85)
        l1 = F.smooth_l1_loss(loc_data, loc_t, reduction='none').sum(2)
        loss_l = (l1 * (weights * pos.float())).sum()

        # hard negatives ranked by objectness loss, 3:1 (:88-96)
        obj_flat = obj_data.reshape(-1, 2)
        obj_lab = obj_t.long().view(-1)
        with torch.no_grad():
            ce = F.cross_entropy(obj_flat, obj_lab, reduction='none')
            ce = ce.masked_fill(obj_t.view(-1), 0.0)
            rank = ce.view(num, -1).sort(1, descending=True)[1].sort(1)[1]
            num_neg = torch.clamp(self.negpos_ratio * num_pos, max=num_priors - 1)
            neg = rank < num_neg.expand_as(rank)
            w = (weights * (pos | neg).float()).view(-1)
        loss_obj = (F.cross_entropy(obj_flat, obj_lab, reduction='none') * w).sum()

        # class loss on objectness-fused logits (:106-117)
        flat_conf = conf_data.reshape(-1, self.num_classes - 1)
        # log(sum(exp(conf))) of :108 as logsumexp: same value, and an unselected row with a huge logit cannot put
        # inf * 0 = NaN into the masked sum
        bg = obj_flat[:, :1] + torch.logsumexp(flat_conf, dim=1, keepdim=True)
        fg = obj_flat[:, 1:2].expand_as(flat_conf) + flat_conf
        logit = torch.cat((bg, fg), 1)
        # ignored boxes carry label -1 (data/voc0712.py:237-238, 263-264: 'incre' phase 2 / instance_shot); the reference
        # never evaluates them (pos is False, their mining loss is zeroed so they are not drawn as negatives, :93) --
        # here their row is evaluated against class 0 and multiplied by w = 0: a valid target index, the same sum
        cls_t = labels.long().clamp_min(0).view(-1)
        loss_c = (F.cross_entropy(logit, cls_t, reduction='none') * w).sum()

        n = num_pos.sum()
        if self.sync_normalizer:          # data-parallel: N over the global batch (see ctdet.dist)
            from ctdet.dist import global_normalizer
            n = global_normalizer(n, dev)
        return {'loss_box_reg': loss_l / n, 'loss_cls': loss_c / n, 'loss_obj': loss_obj / n}
