"""RFBNet-VGG + Context-Transformer on the MI355X-native engine.

Drop-in for the reference's ``models/RFB_Net_vgg.py``: same public names
(``build_net``, ``RFBNet``, ``BasicConv``, ``BasicRFB``, ``BasicRFB_a``), the same
``state_dict`` keys and parameter-name prefixes (``base.``, ``Norm.``, ``extras.``, ``loc.``,
``conf.``, ``obj.``, ``theta/phi/g/Wz/OBJ_Target/scale/fc_base``) so reference checkpoints load
and ``utils/solver.py`` LR groups keep working -- but the modules below are only *parameter
containers*.  ``RFBNet.forward`` hands the whole network to ``ctdet.engine`` which runs it as a
flat sequence of hand-written HIP launches (fp32 convolutions carried on the 16-bit matrix pipe as
split operands -- Winograd F(4x4,3x3) and direct implicit-GEMM kernels, f16x2 or bf16x3 pieces,
csrc/ct_f16x2.h -- pooling, the fused Context-Transformer attention kernel); no ATen convolution,
matmul or softmax is executed.  There is no CPU path: a non-HIP ``device`` raises.

Reference behaviour reproduced (file:line in the reference):
  network topology / layer hyper-parameters      models/RFB_Net_vgg.py:26-112, :323-422
  forward outputs (raw in train, softmax in eval) :273-286;  ``init=True`` early return :250-251
  Context-Transformer block                       :253-271 (phase 2, method 'ours')
  init_weight / normalize                         :297-318
Deliberate differences: the conf head is evaluated once (reference: twice, :240/:243); size 512
with the Context-Transformer uses the build-defined pooling list [3,2,2,2,2,1,1] (the reference
raises IndexError there, :235-244 -- results on it are "parity unpinned").
"""
import math

import os

import torch
import torch.nn as nn

from ctdet import engine as _engine
from ctdet import train_engine as _train
from ctdet import ops as _ops
from ctdet._lib import CtdetError

_VGG_LAYOUT = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'C', 512, 512, 512, 'M', 512, 512, 512)
_EXTRA_RFB = {300: ((1024, 1, 2), (512, 2, 2), (256, 2, 2)),
              512: ((1024, 1, 2), (512, 2, 2), (256, 2, 2), (256, 2, 1), (256, 2, 1))}
_EXTRA_TAIL = {300: ((128, 1, 0), (256, 3, 0), (128, 1, 0), (256, 3, 0)),
               512: ((128, 1, 0), (256, 4, 1))}
_ANCHORS_PER_CELL = {300: (6, 6, 6, 6, 4, 4), 512: (6, 6, 6, 6, 6, 4, 4)}
_CTX_DIMS = {'transfer': (60, 20), 'incre': (15, 5)}


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


class BasicConv(nn.Module):
    """conv (no bias) + BatchNorm2d(eps 1e-5, momentum 0.01) [+ ReLU]; parameters only."""

    def __init__(self, in_planes, out_planes, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 relu=True, bn=True, bias=False):
        super().__init__()
        if groups != 1 or not bn or bias:
            raise CtdetError('BasicConv: only groups=1, bn=True, bias=False are used by RFBNet')
        self.out_channels = out_planes
        self.conv = nn.Conv2d(in_planes, out_planes, kernel_size, stride, padding, dilation, groups, bias)
        self.bn = nn.BatchNorm2d(out_planes, eps=1e-5, momentum=0.01, affine=True)
        self.relu = bool(relu)


def _chain(*specs):
    """specs: (cin, cout, k, stride, pad, dil, relu) -> nn.Sequential of BasicConv."""
    return nn.Sequential(*[BasicConv(ci, co, k, s, p, d, relu=r) for (ci, co, k, s, p, d, r) in specs])


class BasicRFB(nn.Module):
    """Three-branch dilated receptive-field block (reference :26-64)."""

    def __init__(self, in_planes, out_planes, stride=1, scale=0.1, visual=1):
        super().__init__()
        self.scale = scale
        self.out_channels = out_planes
        n = in_planes // 8
        v = visual
        self.branch0 = _chain((in_planes, 2 * n, 1, stride, 0, 1, True),
                              (2 * n, 2 * n, 3, 1, v, v, False))
        self.branch1 = _chain((in_planes, n, 1, 1, 0, 1, True),
                              (n, 2 * n, 3, stride, 1, 1, True),
                              (2 * n, 2 * n, 3, 1, v + 1, v + 1, False))
        self.branch2 = _chain((in_planes, n, 1, 1, 0, 1, True),
                              (n, (n // 2) * 3, 3, 1, 1, 1, True),
                              ((n // 2) * 3, 2 * n, 3, stride, 1, 1, True),
                              (2 * n, 2 * n, 3, 1, 2 * v + 1, 2 * v + 1, False))
        self.ConvLinear = BasicConv(6 * n, out_planes, 1, relu=False)
        self.shortcut = BasicConv(in_planes, out_planes, 1, stride, relu=False)


class BasicRFB_a(nn.Module):
    """RFB-s on conv4_3 (reference :68-112)."""

    def __init__(self, in_planes, out_planes, stride=1, scale=0.1):
        super().__init__()
        if stride != 1:
            raise CtdetError('BasicRFB_a: the engine supports stride 1 (the only use in RFBNet)')
        self.scale = scale
        self.out_channels = out_planes
        n = in_planes // 4
        self.branch0 = _chain((in_planes, n, 1, 1, 0, 1, True),
                              (n, n, 3, 1, 1, 1, False))
        self.branch1 = _chain((in_planes, n, 1, 1, 0, 1, True),
                              (n, n, (3, 1), 1, (1, 0), 1, True),
                              (n, n, 3, 1, 3, 3, False))
        self.branch2 = _chain((in_planes, n, 1, 1, 0, 1, True),
                              (n, n, (1, 3), 1, (0, 1), 1, True),
                              (n, n, 3, 1, 3, 3, False))
        self.branch3 = _chain((in_planes, n // 2, 1, 1, 0, 1, True),
                              (n // 2, (n // 4) * 3, (1, 3), 1, (0, 1), 1, True),
                              ((n // 4) * 3, n, (3, 1), 1, (1, 0), 1, True),
                              (n, n, 3, 1, 5, 5, False))
        self.ConvLinear = BasicConv(4 * n, out_planes, 1, relu=False)
        self.shortcut = BasicConv(in_planes, out_planes, 1, 1, relu=False)


def vgg(cfg=_VGG_LAYOUT, i=3, batch_norm=False):
    """VGG16 trunk as a module list whose indices match the reference (:323-343)."""
    if batch_norm:
        raise CtdetError('vgg(batch_norm=True) is not used by RFBNet')
    mods, cin = [], i
    for v in cfg:
        if v == 'M' or v == 'C':
            mods.append(nn.MaxPool2d(2, 2, ceil_mode=(v == 'C')))
        else:
            mods += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    mods += [nn.MaxPool2d(3, 1, 1),
             nn.Conv2d(512, 1024, 3, padding=6, dilation=6), nn.ReLU(inplace=True),
             nn.Conv2d(1024, 1024, 1), nn.ReLU(inplace=True)]
    return mods


def add_extras(size, cfg=None, in_channels=1024):
    mods, cin = [], in_channels
    for cout, stride, visual in _EXTRA_RFB[size]:
        mods.append(BasicRFB(cin, cout, stride=stride, scale=1.0, visual=visual))
        cin = cout
    for cout, k, pad in _EXTRA_TAIL[size]:
        mods.append(BasicConv(cin, cout, k, 1, pad))
        cin = cout
    return mods


def multibox(size, base_layers, extra_layers, cfg, num_classes):
    indicator = 3 if size == 300 else 5
    src_channels = [512] + [m.out_channels for k, m in enumerate(extra_layers) if k < indicator or k % 2 == 0]
    heads = ([], [], [])
    for ch, anchors in zip(src_channels, cfg):
        for lst, per_anchor in zip(heads, (4, num_classes, 2)):
            lst.append(nn.Conv2d(ch, anchors * per_anchor, 3, padding=1))
    return base_layers, extra_layers, heads


class RFBNet(nn.Module):
    def __init__(self, args, size, base, extras, head, num_classes):
        super().__init__()
        if size not in (300, 512):
            raise CtdetError('only RFBNet300 and RFBNet512 are supported')
        self.method, self.phase, self.setting = args.method, args.phase, args.setting
        self.num_classes = num_classes
        self.size = size
        self.indicator = 3 if size == 300 else 5
        self.base = nn.ModuleList(base)
        self.Norm = BasicRFB_a(512, 512, stride=1, scale=1.0)
        self.extras = nn.ModuleList(extras)
        self.loc, self.conf, self.obj = (nn.ModuleList(h) for h in head)
        self.init_weight()
        if self.method == 'ours' and self.phase == 2:
            d, t = _CTX_DIMS[self.setting]
            if self.setting == 'incre':
                self.fc_base = nn.Linear(d, d)
            self.theta, self.phi, self.g = nn.Linear(d, d), nn.Linear(d, d), nn.Linear(d, d)
            self.Wz = nn.Parameter(torch.zeros(d))
            self.OBJ_Target = nn.Linear(d, t, bias=False)
            self.scale = nn.Parameter(torch.tensor([5.0]), requires_grad=False)
            for lin in (self.theta, self.phi, self.g):
                nn.init.kaiming_normal_(lin.weight, mode='fan_out')
                nn.init.zeros_(lin.bias)
            if self.setting == 'incre':
                nn.init.zeros_(self.fc_base.weight)
                nn.init.zeros_(self.fc_base.bias)
        self._runtimes = {}

    # ------------------------------------------------------------------ parameters
    def init_weight(self):
        """Reference :297-314: kaiming-normal(fan_out) for BasicConv convs, BN gamma=1, every
        bias 0; the plain nn.Conv2d of base/heads keep torch's default weight init."""
        for m in self.modules():
            if isinstance(m, BasicConv):
                nn.init.kaiming_normal_(m.conv.weight, mode='fan_out')
                nn.init.ones_(m.bn.weight)
                nn.init.zeros_(m.bn.bias)
            elif isinstance(m, nn.Conv2d) and m.bias is not None:
                nn.init.zeros_(m.bias)

    def normalize(self):
        """Row-normalise the cosine classifier (reference :316-318)."""
        w = self.OBJ_Target.weight
        self.OBJ_Target.weight.data = w / w.norm(dim=1, keepdim=True)

    def load_weights(self, base_file):
        self.load_state_dict(torch.load(base_file, map_location='cpu'))

    # ------------------------------------------------------------------ execution
    def _device(self):
        dev = getattr(self, 'device', None)
        dev = torch.device(dev) if dev is not None else next(self.parameters()).device
        if dev.type == 'cuda' and dev.index is None:
            dev = torch.device('cuda', torch.cuda.current_device())
        return dev

    def runtime(self, batch, device=None):
        """The engine instance (plan + buffers + packed weights) for this batch size."""
        device = device or self._device()
        if device.type != 'cuda':
            raise CtdetError('RFBNet runs on the MI355X through libctdet only; device %s has no '
                             'implementation (no CPU fallback by design)' % device)
        if next(self.parameters()).device != device:
            raise CtdetError('parameters live on %s but the net was asked to run on %s; call .cuda() '
                             'first' % (next(self.parameters()).device, device))
        dtype = getattr(self, 'conv_dtype', None) or os.environ.get('CTDET_DTYPE', 'f32')
        key = (batch, str(device), dtype)
        if key not in self._runtimes:
            if dtype == 'bf16':          # BASELINE configs[4]: NHWC bf16 activations, bf16 MFMA convolutions
                from ctdet.engine_bf16 import HipBackendBF16
                backend = HipBackendBF16(device)
            elif dtype == 'f32':
                backend = _engine.HipBackend(device)
            else:
                raise CtdetError("conv_dtype must be 'f32' or 'bf16', got %r" % dtype)
            self._runtimes[key] = _engine.Runtime(self, batch, backend)
        return self._runtimes[key]

    def _scale_value(self):
        """float(self.scale) without a device round trip per forward (re-read only when the parameter changed)."""
        key = (self.scale.data_ptr(), self.scale._version)
        if getattr(self, '_scale_cache', (None, None))[0] != key:
            self._scale_cache = (key, float(self.scale.item()))
        return self._scale_cache[1]

    def _ctx_params(self):
        p = dict(theta_w=self.theta.weight, theta_b=self.theta.bias, phi_w=self.phi.weight,
                 phi_b=self.phi.bias, g_w=self.g.weight, g_b=self.g.bias, wz=self.Wz,
                 obj_w=self.OBJ_Target.weight, scale=self._scale_value())
        if self.setting == 'incre':
            p.update(fc_w=self.fc_base.weight, fc_b=self.fc_base.bias)
        return {k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in p.items()}

    def forward_raw(self, x, init=False, _input_loaded=False, _batch=None):
        """-> (loc [B,P,4], conf logits [B,P,C or T], obj logits [B,P,2]) without the eval softmaxes.
        `_input_loaded` (DetectionPipeline): the runtime's input buffer already holds the batch (Runtime.load_input),
        only the launches are issued -- the form a hipGraph captures."""
        if _input_loaded:
            num = _batch
        else:
            x = x.to(self._device(), torch.float32).contiguous()
            num = x.shape[0]
        if self.training:
            # batch-statistics BatchNorm + autograd: the whole backbone is one autograd function
            # whose backward is the HIP backward pass (ctdet.train_engine)
            trt = self.train_runtime(num)
            loc, conf, obj = _train.BackboneFunction.apply(trt, x, not init, *trt.params)
            if init:
                return conf.view(num, -1, self.num_classes)
            if not (self.method == 'ours' and self.phase == 2):
                conf = conf.view(num, -1, self.num_classes)
            return loc.view(num, -1, 4), conf, obj.view(num, -1, 2)
        rt = self.runtime(num)
        loc, conf, obj = rt.run_loaded() if _input_loaded else rt.run_backbone(x)
        conf = conf.view(num, -1, self.num_classes)
        if init:
            return conf
        if self.method == 'ours' and self.phase == 2:
            pool = rt.bufs['pool'].view(num, -1, self.num_classes)
            if 'ctx_out' not in rt.bufs:          # output + workspace owned by the runtime: no allocation per call
                rt.bufs['ctx_out'], rt.bufs['ctx_ws'] = _ops.ctx_attention_buffers(
                    num, conf.shape[1], pool.shape[1], self.num_classes, self.OBJ_Target.weight.shape[0],
                    self.setting == 'incre', conf.device)
            conf = _ops.ctx_attention(conf, pool, self._ctx_params(), self.setting == 'incre',
                                      out=rt.bufs['ctx_out'], ws=rt.bufs['ctx_ws'])
        return loc.view(num, -1, 4), conf, obj.view(num, -1, 2)

    def train_runtime(self, batch, device=None):
        """The training engine (forward with batch-stat BN + HIP backward) for this batch size."""
        device = device or self._device()
        if device.type != 'cuda':
            raise CtdetError('RFBNet trains on the MI355X through libctdet only (device %s)' % device)
        key = ('train', batch, str(device))
        if key not in self._runtimes:
            self._runtimes[key] = _train.TrainRuntime(self, batch, _engine.HipBackend(device))
        return self._runtimes[key]

    def forward(self, x, init=False):
        if self.training:
            return self.forward_raw(x, init)
        with torch.no_grad():
            out = self.forward_raw(x, init)
            if init:
                return out.clone()
            loc, conf, obj = out
            return loc.clone(), _ops.softmax_lastdim(conf), _ops.softmax_lastdim(obj)


def build_net(args, size, num_classes):
    """build_net(args, size in {300,512}, num_classes = #foreground classes of the conf head)."""
    if size not in (300, 512):
        raise CtdetError('only RFBNet300 and RFBNet512 are supported')
    return RFBNet(args, size, *multibox(size, vgg(), add_extras(size), _ANCHORS_PER_CELL[size], num_classes),
                  num_classes)
