"""Batched detection pipeline: the generalisation of test.py:do_test (test.py:121-161) from
one image per step to a whole batch with every stage on the device.

  images [B,3,S,S] -> RFBNet engine (fused HIP convs) -> [Context-Transformer attention]
  -> fused eval-softmax + decode + score fusion + `boxes *= scale`
  -> per (image, class): score > 0.01, descending-score order, NMS 0.45
  -> per image: keep score >= 200-th largest
Results equal the reference's sequential per-image / per-class loop on the same
(loc, conf, obj): all_boxes[img][cls] = float32 [k,5] rows in descending score order.
"""
import torch

from . import ops


class DetectionPipeline:
    def __init__(self, net, priors, batch, num_fg, image_wh=(500, 375), conf_thresh=0.01, nms_thresh=0.45,
                 max_per_image=200, force_cpu_rule=False, variance=(0.1, 0.2), out_cap=None):
        self.net = net
        self.device = net._device()
        self.priors = priors.to(self.device, torch.float32).contiguous()
        self.batch, self.T = batch, num_fg
        self.P = self.priors.shape[0]
        self.variance = variance
        self.conf_thresh, self.nms_thresh, self.max_per_image = conf_thresh, nms_thresh, max_per_image
        self.ge = bool(force_cpu_rule)
        self.set_image_wh(image_wh)
        self.rt = net.runtime(batch, self.device)
        self.post = ops.PostProcessor(batch, self.P, num_fg, self.device, out_cap)
        self.boxes = torch.empty(batch, self.P, 4, device=self.device)
        self.scores = torch.empty(batch, self.P, num_fg + 1, device=self.device)

    def set_image_wh(self, image_wh):
        """`scale` of test.py:122-123: one (w, h) for all images or one per image [B,2]."""
        wh = torch.as_tensor(image_wh, dtype=torch.float32)
        if wh.dim() == 1:
            self.scale = torch.stack([wh[0], wh[1], wh[0], wh[1]]).to(self.device)
        else:
            if wh.shape[0] != self.batch:
                raise ValueError('image_wh has %d rows, pipeline batch is %d' % (wh.shape[0], self.batch))
            self.scale = torch.stack([wh[:, 0], wh[:, 1], wh[:, 0], wh[:, 1]], 1).contiguous().to(self.device)

    @torch.no_grad()
    def run(self, x, image_wh=None):
        """x [B,3,S,S] (device) -> (out_dets [B,T,cap,5], out_count [B,T]) device tensors."""
        if image_wh is not None:
            self.set_image_wh(image_wh)
        loc, conf, obj = self.net.forward_raw(x)
        ops.detect_fused(loc, conf.contiguous(), obj, self.priors, self.variance, True, self.scale,
                         out=(self.boxes, self.scores))
        return self.post.run(self.boxes, self.scores, self.conf_thresh, self.nms_thresh, self.ge,
                             self.max_per_image)

    def results(self):
        """Host copy in the reference's all_boxes layout (synchronises)."""
        return self.post.to_all_boxes()
