"""Batched detection pipeline: the generalisation of test.py:do_test (test.py:121-161) from
one image per step to a whole batch with every stage on the device.

  images [B,3,S,S] -> RFBNet engine (fused HIP convs) -> [Context-Transformer attention]
  -> fused eval-softmax + decode + score fusion + `boxes *= scale`
  -> per (image, class): score > 0.01, descending-score order, NMS 0.45
  -> per image: keep score >= 200-th largest
Results equal the reference's sequential per-image / per-class loop on the same
(loc, conf, obj): all_boxes[img][cls] = float32 [k,5] rows in descending score order.

The whole step (about 100 launches on two streams) allocates nothing and never synchronises with the host, so
after two eager steps it is captured once into a hipGraph and replayed (CTDET_GRAPH=0 keeps eager launches): at
small batches the step is otherwise bound by the host issuing launches, not by the GPU.  Weight re-packing
(parameter updates) stays outside the graph: it rewrites the same packed buffers the captured launches read.
"""
import os
import warnings

import torch

from . import ops


class DetectionPipeline:
    def __init__(self, net, priors, batch, num_fg, image_wh=(500, 375), conf_thresh=0.01, nms_thresh=0.45,
                 max_per_image=200, force_cpu_rule=False, variance=(0.1, 0.2), out_cap=None, graph=None):
        self.net = net
        self.device = net._device()
        self.priors = priors.to(self.device, torch.float32).contiguous()
        self.batch, self.T = batch, num_fg
        self.P = self.priors.shape[0]
        self.variance = variance
        self.conf_thresh, self.nms_thresh, self.max_per_image = conf_thresh, nms_thresh, max_per_image
        self.ge = bool(force_cpu_rule)
        self.scale = None
        self.set_image_wh(image_wh)
        self.rt = net.runtime(batch, self.device)
        self.post = ops.PostProcessor(batch, self.P, num_fg, self.device, out_cap)
        self.boxes = torch.empty(batch, self.P, 4, device=self.device)
        self.scores = torch.empty(batch, self.P, num_fg + 1, device=self.device)
        self.use_graph = (os.environ.get('CTDET_GRAPH', '1') != '0') if graph is None else bool(graph)
        if len(getattr(self.rt, 'sides', None) or []) > 1:
            # more than one side stream (CTDET_STREAMS > 2): ROCm 7.2 crashes in hipStreamEndCapture on that fork/join
            # pattern (measured); those schedules launch eagerly
            self.use_graph = False
        self._graph, self._eager_runs, self._graph_key = None, 0, None

    def set_image_wh(self, image_wh):
        """`scale` of test.py:122-123: one (w, h) for all images or one per image [B,2]."""
        wh = torch.as_tensor(image_wh, dtype=torch.float32)
        if wh.dim() == 1:
            new = torch.stack([wh[0], wh[1], wh[0], wh[1]])
        else:
            if wh.shape[0] != self.batch:
                raise ValueError('image_wh has %d rows, pipeline batch is %d' % (wh.shape[0], self.batch))
            new = torch.stack([wh[:, 0], wh[:, 1], wh[:, 0], wh[:, 1]], 1).contiguous()
        if self.scale is not None and self.scale.shape == new.shape:
            self.scale.copy_(new)                   # same buffer: a captured graph keeps reading it
        else:
            self.scale = new.to(self.device)
            self._graph = None

    def _net_key(self):
        """Everything a captured step froze besides the packed conv weights (which are rewritten in place): whether the
        Context-Transformer block runs, its 'incre' branch, the host-side value of `scale` and the raw pointers of the
        block's parameters -- a load_state_dict that changes `scale`, a re-allocated parameter (net.to(), assign=True)
        or a change of method / phase / setting re-captures instead of replaying stale arguments."""
        net = self.net
        key = (getattr(net, 'method', None), getattr(net, 'phase', None), getattr(net, 'setting', None))
        if key[0] == 'ours' and key[1] == 2:
            prm = net._ctx_params()
            key += tuple(v.data_ptr() if isinstance(v, torch.Tensor) else v for _, v in sorted(prm.items()))
        return key

    def _step(self):
        """The launches of one step on the current stream; reads rt.bufs['x'] (already filled)."""
        loc, conf, obj = self.net.forward_raw(None, _input_loaded=True, _batch=self.batch)
        ops.detect_fused(loc, conf.contiguous(), obj, self.priors, self.variance, True, self.scale,
                         out=(self.boxes, self.scores))
        self.post.run(self.boxes, self.scores, self.conf_thresh, self.nms_thresh, self.ge, self.max_per_image)

    def _capture(self):
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g):
                self._step()
        except Exception as e:                      # not a fallback to another device path: the same HIP launches, eagerly
            warnings.warn('hipGraph capture of the detection step failed (%s: %s); launching eagerly'
                          % (type(e).__name__, e))
            self.use_graph = False
            torch.cuda.synchronize(self.device)
            return None
        return g

    @torch.no_grad()
    def run(self, x, image_wh=None):
        """x [B,3,S,S] (device) -> (out_dets [B,T,cap,5], out_count [B,T]) device tensors."""
        if self.net.training:
            raise RuntimeError('DetectionPipeline runs the eval-mode network (call net.eval() first)')
        if image_wh is not None:
            self.set_image_wh(image_wh)
        rt = self.rt
        rt.load_input(x)                            # shape check, weight refresh, copy into the plan's input buffer
        if hasattr(rt, 'ensure_wired'):
            rt.ensure_wired()
        be = rt.backend
        # (kernel_epoch: a step changed kernels; ws_generation: a three-kernel-Winograd workspace was re-allocated -- either way
        # a captured graph holds stale launches / pointers)
        key = (self.conf_thresh, self.nms_thresh, self.ge, self.max_per_image, rt.event_log is None,
               getattr(be, 'kernel_epoch', 0), getattr(be, 'ws_generation', 0)) + self._net_key()
        if self.use_graph and rt.event_log is None:
            if self._graph is not None and self._graph_key == key:
                self._graph.replay()
                return self.post.out_dets, self.post.out_count
            if self._eager_runs >= 2:               # lazily initialised state (function attributes, autotune) is settled
                self._graph = self._capture()
                self._graph_key = key
                if self._graph is not None:
                    self._graph.replay()
                    return self.post.out_dets, self.post.out_count
        self._eager_runs += 1
        self._step()
        return self.post.out_dets, self.post.out_count

    def results(self):
        """Host copy in the reference's all_boxes layout (synchronises)."""
        return self.post.to_all_boxes()
