"""Inference engine: turns an RFBNet module tree into a flat list of fused HIP launches.

The plan is built once per (network, batch size): every launch owns pre-packed weights,
folded epilogue vectors and pre-allocated activation buffers, so a forward pass is a
straight sequence of `ct_conv2d_fwd` / `ct_maxpool2d_fwd` / ... calls on the current HIP
stream with no allocation and no host synchronisation (and can be captured in a hipGraph).

Fusions relative to the reference's op-by-op PyTorch execution (models/RFB_Net_vgg.py):
  * Conv2d + bias/BatchNorm(eval) + ReLU                      -> one launch (:7-22, :332-336)
  * the parallel 1x1 "reduce" convs of an RFB block + shortcut -> one launch, per-channel ReLU
    mask (:33-50, :75-97)
  * torch.cat of the branches                                  -> branch convs write channel
    slices of one buffer (:58, :106)
  * ConvLinear + `out*scale + short` + ReLU                    -> one launch (:59-62, :107-110)
  * loc/conf/obj head convs + permute + view + cat             -> one launch per source writing
    the flattened [B,P*4] / [B,P*C] / [B,P*2] buffers directly (:238-248)
  * the conf head is evaluated once (the reference runs it twice in phase 2, :240/:243).

The `backend` indirection exists so tests can replay a plan with a torch-CPU emulation of the
launch descriptors (tests/emu_backend.py) and check the wiring without a GPU; the product
ships exactly one backend, the HIP one, and `RFBNet.forward` refuses non-HIP devices.
"""
import ctypes as C
import json
import math
import os
from dataclasses import dataclass, field
from typing import List, Optional

import torch

from . import _lib

# CTDET_TUNE_TABLE: another table file (A/B measurements of a re-tuned table against the committed one)
TUNE_TABLE = os.environ.get('CTDET_TUNE_TABLE') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'conv_tune_gfx950.json')
_tune_table = None


def tune_table():
    """Committed per-shape tile choices measured on an MI355X by tools/tune_convs.py."""
    global _tune_table
    if _tune_table is None:
        try:
            with open(TUNE_TABLE) as f:
                _tune_table = json.load(f)
        except (OSError, ValueError):
            _tune_table = {}
    return _tune_table


CTX_POOL = {300: [3, 2, 2, 2, 1, 1],           # models/RFB_Net_vgg.py:235-236
            512: [3, 2, 2, 2, 2, 1, 1]}        # build-defined: the reference crashes at 512 (:243)


def conv_out(n, k, s, p, d):
    return (n + 2 * p - d * (k - 1) - 1) // s + 1


def pool_out(n, k, s, p, ceil_mode):
    num = n + 2 * p - k
    o = (-(-num // s) if ceil_mode else num // s) + 1
    if ceil_mode and (o - 1) * s >= n + p:
        o -= 1
    return o


@dataclass
class ConvPart:
    """One torch conv folded into a fused launch: weight [cout,cin,kh,kw] + its epilogue source."""
    weight: torch.nn.Parameter
    bias: Optional[torch.nn.Parameter] = None
    bn: Optional[torch.nn.Module] = None
    relu: bool = False

    @property
    def cout(self):
        return self.weight.shape[0]


@dataclass
class Segment:
    dst: str            # flat buffer name [B, img_stride]
    co_begin: int
    co_end: int
    pix_stride: int
    base: int


@dataclass
class ConvStep:
    name: str
    parts: List[ConvPart]
    cin: int
    kh: int
    kw: int
    stride: int
    ph: int
    pw: int
    dil: int
    src: str
    src_coff: int
    h: int
    w: int
    dst: Optional[str] = None
    dst_coff: int = 0
    res: Optional[str] = None
    res_coff: int = 0
    res_scale: float = 1.0
    segs: Optional[List[Segment]] = None
    kind: str = 'conv'
    rt: dict = field(default_factory=dict)      # backend-private runtime state

    @property
    def cout(self):
        return sum(p.cout for p in self.parts)

    @property
    def oh(self):
        return conv_out(self.h, self.kh, self.stride, self.ph, self.dil)

    @property
    def ow(self):
        return conv_out(self.w, self.kw, self.stride, self.pw, self.dil)

    def flops(self, batch):
        return 2.0 * batch * self.cout * self.oh * self.ow * self.cin * self.kh * self.kw

    def tune_key(self, batch):
        return '%dx%d_s%d_d%d_c%d_m%d_%dx%d_b%d%s' % (self.kh, self.kw, self.stride, self.dil, self.cin,
                                                    self.cout, self.h, self.w, batch, '_seg' if self.segs else '')


@dataclass
class PoolStep:
    name: str
    src: str
    dst: str
    ch: int
    h: int
    w: int
    k: int
    stride: int
    pad: int
    ceil_mode: bool
    kind: str = 'pool'

    @property
    def oh(self):
        return pool_out(self.h, self.k, self.stride, self.pad, self.ceil_mode)

    @property
    def ow(self):
        return pool_out(self.w, self.k, self.stride, self.pad, self.ceil_mode)


@dataclass
class CtxPoolStep:
    name: str
    src: str            # flat conf buffer
    src_base: int
    dst: str            # flat pooled buffer
    dst_base: int
    h: int
    w: int
    ch: int
    k: int
    kind: str = 'ctxpool'


class Plan:
    """Flat launch list + buffer table for one (network, batch)."""

    def __init__(self, net, batch):
        self.net = net
        self.batch = batch
        self.size = net.size
        self.C = net.num_classes
        self.ctx = (net.method == 'ours' and net.phase == 2)
        self.steps = []
        self.buf_shapes = {}          # name -> shape tuple (without batch for 4D: (C,H,W); flat: (n,))
        self._build()

    # -- buffer helpers
    def _buf4(self, name, c, h, w):
        self.buf_shapes[name] = (c, h, w)
        return name

    def _flat(self, name, n):
        self.buf_shapes[name] = (n,)
        return name

    def _conv(self, name, parts, src, src_c, h, w, k, stride=1, pad=0, dil=1, src_coff=0, cin=None,
              dst=None, dst_c=None, dst_coff=0, res=None, res_coff=0, res_scale=1.0):
        kh, kw = (k, k) if isinstance(k, int) else k
        ph, pw = (pad, pad) if isinstance(pad, int) else pad
        st = ConvStep(name, parts, cin if cin is not None else src_c, kh, kw, stride, ph, pw, dil,
                      src, src_coff, h, w, dst, dst_coff, res, res_coff, res_scale)
        if dst is not None and dst not in self.buf_shapes:
            self._buf4(dst, dst_c if dst_c is not None else st.cout, st.oh, st.ow)
        self.steps.append(st)
        return st

    @staticmethod
    def _bc(m, relu=None):
        """ConvPart of a BasicConv module (conv without bias + BN [+ ReLU])."""
        return ConvPart(m.conv.weight, None, m.bn, m.relu if relu is None else relu)

    # -- network walk (models/RFB_Net_vgg.py:219-248)
    def _build(self):
        net, S = self.net, self.size
        self._buf4('x', 3, S, S)
        cur, c, h, w = 'x', 3, S, S
        sources = []
        k = 0
        nbase = len(net.base)
        while k < nbase:
            m = net.base[k]
            if isinstance(m, torch.nn.Conv2d):
                relu = k + 1 < nbase and isinstance(net.base[k + 1], torch.nn.ReLU)
                st = self._conv('base.%d' % k, [ConvPart(m.weight, m.bias, None, relu)], cur, c, h, w,
                                m.kernel_size, m.stride[0], m.padding, m.dilation[0], dst='a_base%d' % k)
                cur, c, h, w = st.dst, st.cout, st.oh, st.ow
                k += 2 if relu else 1
                if k == 23:
                    sources.append(self._rfb_a('Norm', net.Norm, cur, c, h, w))
            elif isinstance(m, torch.nn.MaxPool2d):
                ps = PoolStep('base.%d' % k, cur, 'a_base%d' % k, c, h, w, m.kernel_size, m.stride, m.padding,
                              m.ceil_mode)
                self._buf4(ps.dst, c, ps.oh, ps.ow)
                self.steps.append(ps)
                cur, h, w = ps.dst, ps.oh, ps.ow
                k += 1
            else:
                raise TypeError('unexpected module in base: %r' % m)
        indicator = net.indicator
        for i, m in enumerate(net.extras):
            name = 'extras.%d' % i
            if hasattr(m, 'branch0'):
                cur, c, h, w = self._rfb(name, m, cur, c, h, w)
            else:
                st = self._conv(name, [self._bc(m)], cur, c, h, w, m.conv.kernel_size, m.conv.stride[0],
                                m.conv.padding, m.conv.dilation[0], dst='a_' + name)
                cur, c, h, w = st.dst, st.cout, st.oh, st.ow
            if i < indicator or i % 2 == 0:
                sources.append((cur, c, h, w))
        # heads
        mbox = [l.out_channels // 4 for l in net.loc]
        assert len(mbox) == len(sources)
        C = self.C
        self.P = sum(hh * ww * mb for (_, _, hh, ww), mb in zip(sources, mbox))
        self._flat('loc', self.P * 4)
        self._flat('conf', self.P * C)
        self._flat('obj', self.P * 2)
        self.src_info = []
        pbase = 0
        for i, ((sname, sc, sh, sw), mb) in enumerate(zip(sources, mbox)):
            parts = [ConvPart(net.loc[i].weight, net.loc[i].bias), ConvPart(net.conf[i].weight, net.conf[i].bias),
                     ConvPart(net.obj[i].weight, net.obj[i].bias)]
            st = self._conv('head.%d' % i, parts, sname, sc, sh, sw, 3, 1, 1, 1)
            st.segs = [Segment('loc', 0, mb * 4, mb * 4, pbase * 4),
                       Segment('conf', mb * 4, mb * 4 + mb * C, mb * C, pbase * C),
                       Segment('obj', mb * (4 + C), mb * (6 + C), mb * 2, pbase * 2)]
            self.src_info.append((sh, sw, mb, pbase))
            pbase += sh * sw * mb
        if self.ctx:
            kk = CTX_POOL[S]
            assert len(kk) == len(sources)
            self.M = sum(-(-sh // kv) * -(-sw // kv) * mb for (sh, sw, mb, _), kv in zip(self.src_info, kk))
            self._flat('pool', self.M * C)
            mbase = 0
            for i, ((sh, sw, mb, pb), kv) in enumerate(zip(self.src_info, kk)):
                self.steps.append(CtxPoolStep('ctxpool.%d' % i, 'conf', pb * C, 'pool', mbase * C, sh, sw, mb * C, kv))
                mbase += -(-sh // kv) * -(-sw // kv) * mb

    def _rfb_a(self, name, m, src, c, h, w):
        """BasicRFB_a (models/RFB_Net_vgg.py:68-112)."""
        b0, b1, b2, b3 = m.branch0, m.branch1, m.branch2, m.branch3
        i0, i1, i2, i3 = b0[0].out_channels, b1[0].out_channels, b2[0].out_channels, b3[0].out_channels
        co = m.shortcut.out_channels
        t0 = name + '.t0'
        self._conv(name + '.reduce', [self._bc(b0[0]), self._bc(b1[0]), self._bc(b2[0]), self._bc(b3[0]),
                                      self._bc(m.shortcut)], src, c, h, w, 1, dst=t0)
        cat = name + '.cat'
        ccat = b0[1].out_channels + b1[2].out_channels + b2[2].out_channels + b3[3].out_channels
        self._buf4(cat, ccat, h, w)
        off = 0
        self._conv(name + '.b0.1', [self._bc(b0[1])], t0, i0, h, w, 3, 1, 1, 1, src_coff=0, dst=cat, dst_coff=off)
        off += b0[1].out_channels
        self._conv(name + '.b1.1', [self._bc(b1[1])], t0, i1, h, w, (3, 1), 1, (1, 0), 1, src_coff=i0, dst=name + '.t1')
        self._conv(name + '.b1.2', [self._bc(b1[2])], name + '.t1', b1[1].out_channels, h, w, 3, 1, 3, 3, dst=cat, dst_coff=off)
        off += b1[2].out_channels
        self._conv(name + '.b2.1', [self._bc(b2[1])], t0, i2, h, w, (1, 3), 1, (0, 1), 1, src_coff=i0 + i1, dst=name + '.t2')
        self._conv(name + '.b2.2', [self._bc(b2[2])], name + '.t2', b2[1].out_channels, h, w, 3, 1, 3, 3, dst=cat, dst_coff=off)
        off += b2[2].out_channels
        self._conv(name + '.b3.1', [self._bc(b3[1])], t0, i3, h, w, (1, 3), 1, (0, 1), 1, src_coff=i0 + i1 + i2, dst=name + '.t3a')
        self._conv(name + '.b3.2', [self._bc(b3[2])], name + '.t3a', b3[1].out_channels, h, w, (3, 1), 1, (1, 0), 1, dst=name + '.t3b')
        self._conv(name + '.b3.3', [self._bc(b3[3])], name + '.t3b', b3[2].out_channels, h, w, 3, 1, 5, 5, dst=cat, dst_coff=off)
        out = name + '.out'
        # out = relu(ConvLinear(cat)*scale + shortcut(x))   (:107-110)
        self._conv(name + '.linear', [self._bc(m.ConvLinear, relu=True)], cat, ccat, h, w, 1, dst=out,
                   res=t0, res_coff=i0 + i1 + i2 + i3, res_scale=float(m.scale))
        return out, co, h, w

    def _rfb(self, name, m, src, c, h, w):
        """BasicRFB (models/RFB_Net_vgg.py:26-64)."""
        b0, b1, b2 = m.branch0, m.branch1, m.branch2
        s = b0[0].conv.stride[0]
        c0, c1, c2 = b0[0].out_channels, b1[0].out_channels, b2[0].out_channels
        co = m.shortcut.out_channels
        v0, v1, v2 = b0[1].conv.dilation[0], b1[2].conv.dilation[0], b2[3].conv.dilation[0]
        if s == 1:
            t0 = name + '.t0'
            self._conv(name + '.reduce', [self._bc(b0[0]), self._bc(b1[0]), self._bc(b2[0]), self._bc(m.shortcut)],
                       src, c, h, w, 1, dst=t0)
            b0src, b0off, b12src, b1off, b2off = t0, 0, t0, c0, c0 + c1
            shsrc, shoff = t0, c0 + c1 + c2
            ho, wo = h, w
        else:
            ta, tb = name + '.ta', name + '.tb'
            self._conv(name + '.reduce1', [self._bc(b1[0]), self._bc(b2[0])], src, c, h, w, 1, dst=ta)
            st = self._conv(name + '.reduce2', [self._bc(b0[0]), self._bc(m.shortcut)], src, c, h, w, 1, s, dst=tb)
            b0src, b0off, b12src, b1off, b2off = tb, 0, ta, 0, c1
            shsrc, shoff = tb, c0
            ho, wo = st.oh, st.ow
        cat = name + '.cat'
        ccat = b0[1].out_channels + b1[2].out_channels + b2[3].out_channels
        self._buf4(cat, ccat, ho, wo)
        off = 0
        self._conv(name + '.b0.1', [self._bc(b0[1])], b0src, c0, ho, wo, 3, 1, v0, v0, src_coff=b0off, dst=cat, dst_coff=off)
        off += b0[1].out_channels
        self._conv(name + '.b1.1', [self._bc(b1[1])], b12src, c1, h, w, 3, s, 1, 1, src_coff=b1off, dst=name + '.t1')
        self._conv(name + '.b1.2', [self._bc(b1[2])], name + '.t1', b1[1].out_channels, ho, wo, 3, 1, v1, v1, dst=cat, dst_coff=off)
        off += b1[2].out_channels
        self._conv(name + '.b2.1', [self._bc(b2[1])], b12src, c2, h, w, 3, 1, 1, 1, src_coff=b2off, dst=name + '.t2a')
        self._conv(name + '.b2.2', [self._bc(b2[2])], name + '.t2a', b2[1].out_channels, h, w, 3, s, 1, 1, dst=name + '.t2b')
        self._conv(name + '.b2.3', [self._bc(b2[3])], name + '.t2b', b2[2].out_channels, ho, wo, 3, 1, v2, v2, dst=cat, dst_coff=off)
        out = name + '.out'
        self._conv(name + '.linear', [self._bc(m.ConvLinear, relu=True)], cat, ccat, ho, wo, 1, dst=out,
                   res=shsrc, res_coff=shoff, res_scale=float(m.scale))
        return out, co, ho, wo

    def conv_flops(self):
        return sum(s.flops(self.batch) for s in self.steps if s.kind == 'conv')


# ======================================================================================
WINO = -1          # ConvStep.rt['config'] value selecting the Winograd F(2x2,3x3) kernel
WINO4 = -2         # ... the Winograd F(4x4,3x3) kernel
WINOX = -3         # ... F(2x2,3x3) on the bf16 matrix pipe (bf16x3), two accumulators (csrc/ct_wino_x3.hip)
WINO4S = -6        # ... F(4x4,3x3) as transform / bf16x3 GEMM / transform kernels, two accumulators (csrc/ct_wino4s.hip)
WINO4F = -8        # ... F(4x4,3x3) fused on bf16x3, one 64-cout block per workgroup (csrc/ct_wino4f.hip): the narrow layers on big maps
WINO4H = -9        # ... the three-kernel form on the f16x2 operand form (two binary16 pieces, three products; csrc/ct_f16x2.h)
WINO4FH = -10      # ... the fused kernel on the f16x2 operand form
# st.rt['wino'] values: 2, 4 = the fp32-MFMA kernels' tile sizes; 23 = F(2x2,3x3) on bf16x3 (two accumulators, eight waves)
# 44 = F(4x4,3x3) in the three-kernel form with the GEMMs on bf16x3 (two accumulators); 46 = F(4x4,3x3) fused on bf16x3
# (codes 24 and 45, the one-accumulator variants of 23 and 44, existed in rounds 4-5: never selected, removed in round 6)
# 47 = the three-kernel form with its GEMMs on f16x2 (two binary16 pieces, three products, two accumulators)
# 48 = the fused F(4x4,3x3) kernel on f16x2
WINO_TILE = {WINO: 2, WINO4: 4, WINOX: 23, WINO4S: 44, WINO4F: 46, WINO4H: 47, WINO4FH: 48}
WINO_NAME = {2: 'wino', 4: 'wino4', 23: 'winox', 44: 'wino4s', 46: 'wino4f', 47: 'wino4h', 48: 'wino4fh'}
WINOX_TILES = (23,)
WINOX_VARIANT = {23: 1}               # the `variant` argument of ct_conv2d_wino_x3_fwd
WINO4S_TILES = (44, 47)
WINO4S_VARIANT = {44: 1, 47: 3}       # the `variant` argument of ct_conv2d_wino4s_fwd
WINO4H_TILES = (47,)                         # ... whose weights come from ct_conv_pack_weights_wino4s_h2
WINO4F_TILES = (46, 48)
WINO4F_VARIANT = {46: 1, 48: 2}              # the `variant` argument of ct_conv2d_wino4f_pool_fwd_v
WINO4FH_TILES = (48,)                        # ... whose weights come from ct_conv_pack_weights_wino4f_h2 and which needs desc.in_absmax
H2_TILES = (47, 48)                          # the f16x2 operand form: consumers of a maximum of |input| (ct_conv_desc.in_absmax)
TRACK_TILES = (44, 47, 48)               # kernels that fold max |output| into ct_conv_desc.out_absmax (+ the 'valu' image layer)
F4_TILES = (4, 44, 46, 47, 48)           # every variant with F(4x4,3x3)'s rounding (accuracy policies treat them alike)


class HipBackend:
    """Executes plan steps through libctdet (the only backend the product ships)."""

    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise _lib.CtdetError('HipBackend needs a HIP device, got %s' % device)
        self.lib = _lib.lib()
        # V / M workspace of the three-kernel Winograd form (csrc/ct_wino4s.hip): ONE buffer per stream of the schedule,
        # sized for the largest layer on it -- launches of a stream run one after another (st.rt['ws_key'] = the stream
        # index the Runtime gave the step, 0 before there is a schedule).  Per-layer buffers were ~6 GB per RFBNet-300
        # runtime at bs 32 and > 10 GB for RFBNet-512.
        self.ws_pool = {}
        self.ws_generation = 0
        # "slots" for the per-image maxima of |activation| (ct_conv_desc.in_absmax / out_absmax, include/ctdet.h: one line per
        # image): rows of one tensor that a runtime zeroes once per step (zero_slots)
        self.slot_pool = None
        self.slots_used = 0
        self.kernel_epoch = 0          # bumped whenever a step changes kernels (enable_wino / enable_x3): a Runtime re-wires its slots

    def new_slot(self, batch):
        """Device pointer of a fresh slot of `batch` lines of CT_ABSMAX_LINE_BYTES."""
        words = batch * _lib.ABSMAX_LINE_BYTES // 4
        if self.slot_pool is None or self.slot_pool.shape[1] < words:
            if self.slots_used:
                raise _lib.CtdetError('absmax slots were handed out for a smaller batch')
            self.slot_pool = torch.zeros((128, words), device=self.device, dtype=torch.int32)
        if self.slots_used >= self.slot_pool.shape[0]:
            raise _lib.CtdetError('out of absmax slots')
        self.slots_used += 1
        return self.slot_pool[self.slots_used - 1].data_ptr()

    def zero_slots(self):
        if self.slot_pool is not None and self.slots_used:
            self.slot_pool[:self.slots_used].zero_()

    def _ws_drop(self, key):
        """Give a workspace back.  Launches on the side streams may still be using it (it was allocated on the caller's stream,
        so the caching allocator would hand the block out again at once), and a captured hipGraph holds its raw pointer: wait
        for the device, and bump the generation a DetectionPipeline keys its graph on."""
        if self.ws_pool.get(key) is not None:
            torch.cuda.synchronize(self.device)
            self.ws_generation += 1
        self.ws_pool[key] = None

    def ws_reserve(self, key, nbytes):
        t = self.ws_pool.get(key)
        if t is None or t.numel() < nbytes:
            self._ws_drop(key)                  # drop the old buffer before allocating the larger one
            self.ws_pool[key] = self.alloc((nbytes,), torch.uint8)

    def ws_rebuild(self, steps):
        """(Re)size the pool to what the given conv steps need, per key; called once the schedule is known."""
        need = {}
        for st in steps:
            if st.rt.get('wino') in WINO4S_TILES:
                k = st.rt.get('ws_key', 0)
                need[k] = max(need.get(k, 0), st.rt.get('ws4s_bytes', 0))
        for k in list(self.ws_pool):
            if k not in need or self.ws_pool[k] is None or self.ws_pool[k].numel() != need[k]:
                self._ws_drop(k)
        for k, n in need.items():
            if self.ws_pool.get(k) is None:
                self.ws_pool[k] = self.alloc((n,), torch.uint8)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def alloc(self, shape, dtype=torch.float32):
        return torch.empty(shape, device=self.device, dtype=dtype)

    # ---- conv
    def prepare_conv(self, st, bufs, batch):
        lib = self.lib
        kpad = lib.ct_conv_kpad(st.cin, st.kh, st.kw)
        if kpad < 0:
            raise _lib.CtdetError('%s: %dx%d filters are not built' % (st.name, st.kh, st.kw))
        mpad = lib.ct_conv_mpad(st.cout)
        rt = st.rt
        rt['wpk'] = self.alloc((kpad, mpad))
        rt['scale'] = torch.ones(mpad, device=self.device)       # entries past cout stay (1, 0)
        rt['shift'] = torch.zeros(mpad, device=self.device)
        relus = [p.relu for p in st.parts]
        rt['lo'] = None
        if any(relus) and not all(relus):
            lo = torch.empty(mpad, device=self.device)
            off = 0
            for p in st.parts:
                lo[off:off + p.cout] = 0.0 if p.relu else -float('inf')
                off += p.cout
            lo[off:] = 0.0
            rt['lo'] = lo
        rt['kpad'], rt['mpad'] = kpad, mpad
        self.pack_conv(st)
        d = _lib.ConvDesc()
        src = bufs[st.src]
        d.in_ = src.data_ptr()
        d.batch, d.cin, d.h, d.w = batch, st.cin, st.h, st.w
        d.in_ctot, d.in_coff = src.shape[1], st.src_coff
        d.wpacked, d.scale, d.shift = rt['wpk'].data_ptr(), rt['scale'].data_ptr(), rt['shift'].data_ptr()
        d.cout, d.m_pad, d.k_pad = st.cout, mpad, kpad
        d.kh, d.kw, d.stride, d.pad_h, d.pad_w, d.dil = st.kh, st.kw, st.stride, st.ph, st.pw, st.dil
        d.oh, d.ow = st.oh, st.ow
        if st.segs:
            d.nseg = len(st.segs)
            for g, sg in enumerate(st.segs):
                t = bufs[sg.dst]
                d.seg[g].ptr = t.data_ptr()
                d.seg[g].co_begin, d.seg[g].co_end = sg.co_begin, sg.co_end
                d.seg[g].pix_stride = sg.pix_stride
                d.seg[g].img_stride = t.shape[1]
                d.seg[g].base = sg.base
        else:
            dst = bufs[st.dst]
            assert dst.shape[2] == st.oh and dst.shape[3] == st.ow, (st.name, dst.shape, st.oh, st.ow)
            d.out, d.out_ctot, d.out_coff = dst.data_ptr(), dst.shape[1], st.dst_coff
        if st.res is not None:
            r = bufs[st.res]
            d.res, d.res_ctot, d.res_coff, d.res_scale = r.data_ptr(), r.shape[1], st.res_coff, st.res_scale
        d.relu = int(all(relus))
        d.lo = rt['lo'].data_ptr() if rt['lo'] is not None else None
        d.config = max(rt.get('config', 0), 0)
        # split-K workspace for maps that cannot fill the chip (the library decides per launch whether to use it)
        npix = batch * st.oh * st.ow
        if int(os.environ.get('CTDET_KSPLIT', '1')):
            # 16 slabs for the small maps; 4 for the mid-sized 3x3 layers, where only the F(4x4,3x3) kernel splits
            # (over input channels: a 32-tile x 64-cout workgroup grid is 16x coarser than the direct kernel's)
            slabs = 16 if st.cout * npix <= (2 << 20) else \
                4 if st.cout * npix <= (8 << 20) and (st.kh, st.kw, st.stride, st.dil) == (3, 3, 1, 1) else 0
            force = int(os.environ.get('CTDET_FORCE_KSPLIT', '0'))      # accuracy experiments (tools/ctx_parity.py)
            if force > 1:
                slabs = force
            if slabs:
                rt['ksws'] = torch.empty(slabs * st.cout * npix, device=self.device)
                d.ksplit, d.ksplit_ws, d.ksplit_ws_floats = (force if force > 1 else -1), rt['ksws'].data_ptr(), \
                    rt['ksws'].numel()
        rt['desc'] = d
        rt['wino_ok'] = bool(lib.ct_conv_wino_supported(C.byref(d)))
        rt['winox_ok'] = bool(lib.ct_conv_wino_x3_supported(C.byref(d)))
        rt['wino4s_ok'] = bool(lib.ct_conv_wino4s_supported(C.byref(d)))
        rt['wino4f_ok'] = bool(lib.ct_conv_wino4f_supported(C.byref(d)))
        if rt.get('config', 0) in WINO_TILE:
            self.enable_wino(st, tile=WINO_TILE[rt['config']])

    def enable_wino(self, st, on=True, tile=None):
        """Route this conv through a Winograd kernel (3x3 s1 d1 p1 layers only): tile 2 = F(2x2,3x3),
        tile 4 = F(4x4,3x3), 23 = F(2x2,3x3) on the bf16 matrix pipe (cin % 16 == 0), 44 / 46 / 47 / 48 = the F(4x4,3x3) forms on the
        16-bit pipe.  st.rt['wino'] holds the code in use."""
        rt = st.rt
        self.kernel_epoch += 1
        if not on:
            rt['wino'] = False
            rt.pop('ws4s_bytes', None)      # the three-kernel form's V / M workspace need (ws_rebuild shrinks the pool)
            if rt.get('wpk_stale'):
                self.pack_conv(st)
            return
        tile = int(tile or 2)
        # the three-kernel form also takes dilated 3x3 layers (pad = dilation: tiles on the sub-lattices), which the fused
        # kernels do not
        if not (rt.get('wino_ok') or (tile in WINO4S_TILES and rt.get('wino4s_ok'))):
            raise _lib.CtdetError('%s: geometry has no Winograd path' % st.name)
        rt['x3'] = None
        if tile not in (2, 4) + WINOX_TILES + WINO4S_TILES + WINO4F_TILES:
            raise _lib.CtdetError('%s: Winograd tile %r (2, 4, 23, 44, 46, 47 or 48)' % (st.name, tile))
        if tile not in WINO4S_TILES:
            rt.pop('ws4s_bytes', None)
        if tile in WINO4S_TILES:
            if not rt.get('wino4s_ok'):
                raise _lib.CtdetError('%s: geometry has no three-kernel Winograd path (cin %% 16)' % st.name)
            if tile in WINO4H_TILES:
                if 'U4H' not in rt:
                    rt['U4H'] = self.alloc((self.lib.ct_conv_wino4s_h2_packed_bytes(st.cin, st.cout),), torch.uint8)
            elif 'U4S' not in rt:
                rt['U4S'] = self.alloc((self.lib.ct_conv_wino4s_packed_bytes(st.cin, st.cout),), torch.uint8)
            rt['ws4s_bytes'] = self.lib.ct_conv_wino4s_workspace_bytes(C.byref(rt['desc']))
            self.ws_reserve(rt.get('ws_key', 0), rt['ws4s_bytes'])
        elif tile in WINO4F_TILES:
            if not rt.get('wino4f_ok'):
                raise _lib.CtdetError('%s: geometry has no fused F(4x4,3x3) bf16x3 path (cin %% 16)' % st.name)
            if tile in WINO4FH_TILES:
                if 'U4FH' not in rt:
                    rt['U4FH'] = self.alloc((self.lib.ct_conv_wino4f_h2_packed_bytes(st.cin, st.cout),), torch.uint8)
            elif 'U4F' not in rt:
                rt['U4F'] = self.alloc((self.lib.ct_conv_wino4f_packed_bytes(st.cin, st.cout),), torch.uint8)
        elif tile in WINOX_TILES:
            if not rt.get('winox_ok'):
                raise _lib.CtdetError('%s: geometry has no Winograd bf16x3 path (cin %% 16)' % st.name)
            if 'UX' not in rt:
                rt['UX'] = self.alloc((self.lib.ct_conv_wino_x3_packed_bytes(st.cin, st.cout),), torch.uint8)
        else:
            key = 'U' if tile == 2 else 'U4'
            if key not in rt:
                sizeof = self.lib.ct_conv_wino_packed_floats if tile == 2 else self.lib.ct_conv_wino4_packed_floats
                rt[key] = self.alloc((sizeof(st.cin, st.cout),))
        rt['wino'] = tile
        self._pack_wino(st)

    def _pack_wino(self, st):
        n = len(st.parts)
        ptrs = (C.c_void_p * n)(*[p.weight.detach().data_ptr() for p in st.parts])
        couts = (C.c_int * n)(*[p.cout for p in st.parts])
        if st.rt['wino'] == 4:
            _lib.check(self.lib.ct_conv_pack_weights_wino4(ptrs, couts, n, st.cin, st.rt['U4'].data_ptr(), self._stream()),
                       'ct_conv_pack_weights_wino4')
            return
        if st.rt['wino'] in WINOX_TILES:
            _lib.check(self.lib.ct_conv_pack_weights_wino_x3(ptrs, couts, n, st.cin, st.rt['UX'].data_ptr(), self._stream()),
                       'ct_conv_pack_weights_wino_x3')
            return
        if st.rt['wino'] in WINO4H_TILES:
            _lib.check(self.lib.ct_conv_pack_weights_wino4s_h2(ptrs, couts, n, st.cin, st.rt['U4H'].data_ptr(), self._stream()),
                       'ct_conv_pack_weights_wino4s_h2')
            return
        if st.rt['wino'] in WINO4S_TILES:
            _lib.check(self.lib.ct_conv_pack_weights_wino4s(ptrs, couts, n, st.cin, st.rt['U4S'].data_ptr(), self._stream()),
                       'ct_conv_pack_weights_wino4s')
            return
        if st.rt['wino'] in WINO4FH_TILES:
            _lib.check(self.lib.ct_conv_pack_weights_wino4f_h2(ptrs, couts, n, st.cin, st.rt['U4FH'].data_ptr(), self._stream()),
                       'ct_conv_pack_weights_wino4f_h2')
            return
        if st.rt['wino'] in WINO4F_TILES:
            _lib.check(self.lib.ct_conv_pack_weights_wino4f(ptrs, couts, n, st.cin, st.rt['U4F'].data_ptr(), self._stream()),
                       'ct_conv_pack_weights_wino4f')
            return
        _lib.check(self.lib.ct_conv_pack_weights_wino(ptrs, couts, n, st.cin, st.rt['U'].data_ptr(), self._stream()),
                   'ct_conv_pack_weights_wino')

    def pack_conv(self, st, weights=True):
        """(Re)pack weights and fold the epilogue from the CURRENT parameter values.  weights=False: only the epilogue
        (the training engine re-packs all weights of a step in one batched launch, ct_pack_run)."""
        rt, lib = st.rt, self.lib
        n = len(st.parts)
        ws = [p.weight.detach() for p in st.parts]
        for wt in ws:
            if not (wt.is_cuda and wt.is_contiguous() and wt.dtype == torch.float32):
                raise _lib.CtdetError('%s: parameters must be contiguous fp32 on the HIP device' % st.name)
        ptrs = (C.c_void_p * n)(*[wt.data_ptr() for wt in ws])
        couts = (C.c_int * n)(*[p.cout for p in st.parts])
        if not weights:
            pass
        elif rt.get('wino'):                    # only the layout the launch reads; the other one is packed on demand
            self._pack_wino(st)
            rt['wpk_stale'] = True
        elif rt.get('x3') is not None:
            self._pack_x3(st)
            rt['wpk_stale'] = True
        else:
            _lib.check(lib.ct_conv_pack_weights(ptrs, couts, n, st.cin, st.kh, st.kw, rt['wpk'].data_ptr(),
                                                rt['mpad'], rt['kpad'], self._stream()), 'ct_conv_pack_weights')
            rt['wpk_stale'] = False
        first = not rt.get('folded')
        rt['folded'] = True
        off = 0
        for p in st.parts:
            if not first and p.bn is None and p.bias is None:
                off += p.cout                   # identity epilogue: constant, written by the first call
                continue
            if p.bn is not None:
                bn = p.bn
                args = (bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                        bn.running_var.data_ptr(), float(bn.eps), None)
            else:
                args = (None, None, None, None, 0.0, p.bias.data_ptr() if p.bias is not None else None)
            _lib.check(lib.ct_conv_fold_epilogue(*args, p.cout, off, rt['scale'].data_ptr(),
                                                 rt['shift'].data_ptr(), self._stream()), 'ct_conv_fold_epilogue')
            off += p.cout
        rt['versions'] = self.param_versions(st)

    @staticmethod
    def param_versions(st):
        v = []
        for p in st.parts:
            v.append((p.weight.data_ptr(), p.weight._version))
            if p.bias is not None:
                v.append((p.bias.data_ptr(), p.bias._version))
            if p.bn is not None:
                # num_batches_tracked stands in for the running statistics: the training engine updates
                # running_mean / running_var through raw pointers (no version bump) but bumps this counter
                # once per training forward, so an eval runtime re-folds its epilogue after any train step
                for t in (p.bn.weight, p.bn.bias, p.bn.running_mean, p.bn.running_var, p.bn.num_batches_tracked):
                    if t is not None:
                        v.append((t.data_ptr(), t._version))
        return v

    def _own_absmax(self, st):
        """A consumer of ct_conv_desc.in_absmax whose input has no producer-side maximum (a layer run on its own, an input some
        other kernel wrote): a private slot, zeroed and filled by ct_absmax_f32 in front of the launch."""
        rt, d = st.rt, st.rt['desc']
        if rt.get('amax_own') is None:
            rt['amax_own'] = torch.zeros(d.batch * _lib.ABSMAX_LINE_BYTES // 4, device=self.device, dtype=torch.int32)
            d.in_absmax = rt['amax_own'].data_ptr()
        rt['amax_own'].zero_()
        hw = d.h * d.w
        base = d.in_ + 4 * d.in_coff * hw
        _lib.check(self.lib.ct_absmax_f32(base, d.batch, d.cin * hw, d.in_ctot * hw, rt['amax_own'].data_ptr(), self._stream()),
                   'ct_absmax_f32')

    def run_conv(self, st):
        tile = st.rt.get('wino')
        needs_max = tile in WINO4FH_TILES or (not tile and st.rt.get('x3') is not None and self.x3_h2(st.rt['x3']))
        if needs_max and (not st.rt['desc'].in_absmax or (st.rt.get('amax_own') is not None and not st.rt.get('amax_frozen'))):
            self._own_absmax(st)
        if tile in WINO4S_TILES:         # F(4x4,3x3): transform / bf16x3 GEMM / transform (csrc/ct_wino4s.hip)
            lib, U, ws, var = self.lib, st.rt['U4H' if tile in WINO4H_TILES else 'U4S'].data_ptr(), \
                self.ws_pool[st.rt.get('ws_key', 0)], WINO4S_VARIANT[tile]
            pool = st.rt.get('pool')
            if pool is not None:
                t, poh, pow_, full = pool
                _lib.check(lib.ct_conv2d_wino4s_pool_fwd(C.byref(st.rt['desc']), U, ws.data_ptr(), ws.numel(), var,
                                                         t.data_ptr(), t.shape[1], 0, poh, pow_, int(full), self._stream()),
                           st.name)
                return
            _lib.check(lib.ct_conv2d_wino4s_fwd(C.byref(st.rt['desc']), U, ws.data_ptr(), ws.numel(), var, self._stream()),
                       st.name)
            return
        if tile in WINO4F_TILES:         # F(4x4,3x3) fused on the bf16 matrix pipe (csrc/ct_wino4f.hip)
            lib, U, var = self.lib, st.rt['U4FH' if tile in WINO4FH_TILES else 'U4F'].data_ptr(), WINO4F_VARIANT[tile]
            pool = st.rt.get('pool')
            if pool is not None:
                t, poh, pow_, full = pool
                _lib.check(lib.ct_conv2d_wino4f_pool_fwd_v(C.byref(st.rt['desc']), U, var, t.data_ptr(), t.shape[1], 0, poh, pow_,
                                                           int(full), self._stream()), st.name)
                return
            _lib.check(lib.ct_conv2d_wino4f_pool_fwd_v(C.byref(st.rt['desc']), U, var, None, 0, 0, 0, 0, 1, self._stream()), st.name)
            return
        if tile in WINOX_TILES:          # F(2x2,3x3) on the bf16 matrix pipe (csrc/ct_wino_x3.hip)
            lib, U, dual = self.lib, st.rt['UX'].data_ptr(), WINOX_VARIANT[tile]
            pool = st.rt.get('pool')
            if pool is not None:
                t, poh, pow_, full = pool
                _lib.check(lib.ct_conv2d_wino_x3_pool_fwd(C.byref(st.rt['desc']), U, dual, t.data_ptr(), t.shape[1], 0, poh,
                                                          pow_, int(full), self._stream()), st.name)
                return
            _lib.check(lib.ct_conv2d_wino_x3_fwd(C.byref(st.rt['desc']), U, dual, self._stream()), st.name)
            return
        if tile:
            lib = self.lib
            U = st.rt['U4' if tile == 4 else 'U'].data_ptr()
            pool = st.rt.get('pool')
            if pool is not None:        # fused MaxPool2d(2, 2): (pooled buffer, oh, ow, write_full)
                t, poh, pow_, full = pool
                fn = lib.ct_conv2d_wino4_pool_fwd if tile == 4 else lib.ct_conv2d_wino_pool_fwd
                _lib.check(fn(C.byref(st.rt['desc']), U, t.data_ptr(), t.shape[1], 0, poh, pow_, int(full),
                              self._stream()), st.name)
                return
            fn = lib.ct_conv2d_wino4_fwd if tile == 4 else lib.ct_conv2d_wino_fwd
            _lib.check(fn(C.byref(st.rt['desc']), U, self._stream()), st.name)
            return
        x3 = st.rt.get('x3')
        if x3 is not None:               # fp32 convolution on the bf16 matrix pipe (bf16x3 split, csrc/ct_conv_x3.hip)
            _lib.check(self.lib.ct_conv2d_x3_fwd(C.byref(st.rt['desc']), st.rt['wx3'][(self.x3_bk(x3), self.x3_h2(x3))].data_ptr(), x3,
                                                 self._stream()), st.name)
            return
        _lib.check(self.lib.ct_conv2d_fwd(C.byref(st.rt['desc']), self._stream()), st.name)

    # ---- bf16x3 (ct_conv2d_x3_fwd): config index, or None for the fp32 MFMA kernel
    def x3_bk(self, cfg):
        return self.lib.ct_conv_x3_config_bk(cfg)

    def x3_h2(self, cfg):
        """Whether bf16x3-kernel config `cfg` is one of the f16x2 twins ('h2:<tile>', csrc/ct_f16x2.h)."""
        return bool(self.lib.ct_conv_x3_config_h2(cfg))

    def x3_names(self):
        return [self.lib.ct_conv_x3_config_name(i).decode() for i in range(self.lib.ct_conv_x3_num_configs())]

    def enable_x3(self, st, cfg):
        """Route this conv through the bf16x3 kernel with tile config `cfg` (None = back to ct_conv2d_fwd)."""
        rt = st.rt
        self.kernel_epoch += 1
        if cfg is None:
            rt['x3'] = None
            if rt.get('wpk_stale'):
                self.pack_conv(st)
            return
        self.enable_wino(st, False)
        bk = self.x3_bk(cfg)
        if bk <= 0:
            raise _lib.CtdetError('%s: bf16x3 config %r' % (st.name, cfg))
        rt.setdefault('wx3', {})
        h2 = self.x3_h2(cfg)
        if (bk, h2) not in rt['wx3']:
            size = self.lib.ct_conv_x3h_packed_bytes if h2 else self.lib.ct_conv_x3_packed_bytes
            rt['wx3'][(bk, h2)] = self.alloc((size(st.cin, st.cout, st.kh, st.kw, bk),), torch.uint8)
        rt['x3'] = cfg
        self._pack_x3(st)

    def _pack_x3(self, st):
        n = len(st.parts)
        ptrs = (C.c_void_p * n)(*[p.weight.detach().data_ptr() for p in st.parts])
        couts = (C.c_int * n)(*[p.cout for p in st.parts])
        bk, h2 = self.x3_bk(st.rt['x3']), self.x3_h2(st.rt['x3'])
        pack = self.lib.ct_conv_pack_weights_x3h if h2 else self.lib.ct_conv_pack_weights_x3
        _lib.check(pack(ptrs, couts, n, st.cin, st.kh, st.kw, bk, st.rt['wx3'][(bk, h2)].data_ptr(), self._stream()),
                   'ct_conv_pack_weights_x3h' if h2 else 'ct_conv_pack_weights_x3')

    def run_pool(self, st, bufs, batch):
        _lib.check(self.lib.ct_maxpool2d_fwd(bufs[st.src].data_ptr(), bufs[st.dst].data_ptr(), batch * st.ch,
                                             st.h, st.w, st.oh, st.ow, st.k, st.stride, st.pad, self._stream()),
                   st.name)

    def run_ctxpool(self, st, bufs, batch):
        src, dst = bufs[st.src], bufs[st.dst]
        _lib.check(self.lib.ct_ctx_pool_fwd(src.data_ptr() + 4 * st.src_base, src.shape[1],
                                            dst.data_ptr() + 4 * st.dst_base, dst.shape[1], batch, st.h, st.w,
                                            st.ch, st.k, self._stream()), st.name)

    # ---- autotune: time every tile config of a conv step on its real buffers
    def _time_conv(self, st, iters=3, rounds=2):
        """ms per launch: the better of `rounds` timed bursts of `iters` launches after one warm-up launch.  Short
        launches get longer bursts (>= ~1 ms of device time): three 30 us launches are inside the timer's noise, and a
        flipped choice lands in the committed table.  A kernel on the f16x2 operand form is timed with the maxima of |input| in
        place (taken once here, as its producer would have left them in a network), not with an absmax pass per launch."""
        tile = st.rt.get('wino')
        if tile in H2_TILES or (not tile and st.rt.get('x3') is not None and self.x3_h2(st.rt['x3'])):
            if not st.rt['desc'].in_absmax or st.rt.get('amax_own') is not None:
                st.rt['amax_frozen'] = False
                self._own_absmax(st)
                st.rt['amax_frozen'] = True
        self.run_conv(st)
        torch.cuda.synchronize(self.device)
        best = float('inf')
        for r in range(rounds + 1):
            if r == 1 and best < 1.0 / 3:             # the first burst was the estimate: size the others from it
                iters = min(64, max(iters, int(1.0 / max(best, 1e-3)) + 1))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                self.run_conv(st)
            e1.record()
            torch.cuda.synchronize(self.device)
            best = min(best, e0.elapsed_time(e1) / iters)
        return best

    def tune_conv(self, st, iters=3):
        ncfg = self.lib.ct_conv_num_configs()
        best, best_t = 0, float('inf')
        times = []
        self.enable_wino(st, False)
        self.enable_x3(st, None)
        for cfg in range(ncfg):
            st.rt['desc'].config = cfg + 1
            try:
                t = self._time_conv(st, iters)
            except _lib.CtdetError:
                times.append(float('inf'))
                continue
            times.append(t)
            if t < best_t:
                best, best_t = cfg, t
        st.rt['desc'].config = best + 1
        st.rt['config'] = best + 1
        best_x3 = None
        if x3_allowed(st):
            for cfg in range(self.lib.ct_conv_x3_num_configs()):
                if st.cin % self.x3_bk(cfg) or not self.x3_names()[cfg].endswith('d') or self.x3_h2(cfg):   # (the table holds the bf16x3 names)
                    times.append(float('inf'))      # k-step does not divide cin / single accumulator (accuracy gate)
                    continue
                self.enable_x3(st, cfg)
                t = self._time_conv(st, iters)
                times.append(t)
                if t < best_t:
                    best_x3, best_t = cfg, t
            self.enable_x3(st, best_x3)
        if (st.rt.get('wino_ok') or st.rt.get('wino4s_ok')) and os.environ.get('CTDET_WINO', '1') != '0':
            best_tile = 0
            for tile in wino_tiles(self, st):
                self.enable_wino(st, tile=tile)
                t = self._time_conv(st, iters)
                times.append(t)
                if t < best_t:
                    best_tile, best_t = tile, t
            # what the layer would run without a Winograd kernel: recorded next to a dilated layer's 'wino4s' entry ('|alt')
            st.rt['alt'] = self.x3_names()[best_x3] if best_x3 is not None else self.lib.ct_conv_config_name(best).decode()
            if best_tile:
                self.enable_wino(st, tile=best_tile)
            else:
                self.enable_wino(st, False)
                if best_x3 is not None:
                    self.enable_x3(st, best_x3)
        st.rt['tune_ms'] = times
        return best, times


def x3_allowed(st):
    """bf16x3 (ct_conv2d_x3_fwd) is a candidate for every forward conv except the 3-channel image layer (K = 27:
    nothing to gain); CTDET_X3=0 keeps the fp32 MFMA kernel everywhere."""
    return os.environ.get('CTDET_X3', '1') != '0' and st.cin >= 16 and st.cin % 16 == 0


def wino_tiles(backend=None, st=None):
    """Winograd variants the tuner / the table may use (st.rt['wino'] codes).  Default '2,4,44,46': the two fp32-MFMA kernels,
    the three-kernel F(4x4,3x3) / bf16x3 form with two accumulators (csrc/ct_wino4s.hip) and the fused F(4x4,3x3) / bf16x3
    kernel for the narrow layers on big maps (csrc/ct_wino4f.hip).  The fused F(2x2) bf16x3
    form (23) is faster than the fused fp32 kernels per layer ALONE (512 -> 512 @38x38: 742 -> 630 us) but not in the
    two-stream pipeline (same-box A/B of two tables: 3 360-3 371 vs 3 228-3 398 images/s) and slower
    than tile 44 wherever that applies, so the committed table does not hold it; CTDET_WINO_TILES=2,4,23,44 lets the tuner
    time it.  A runtime with an accuracy policy
    (ctx_tile_set: networks with the Context-Transformer block) uses ITS set instead (narrowed by an explicit
    CTDET_WINO_TILES), plus F(4x4) / fp32 on layers with at most wino4_max_cin input channels (and tile 44 from
    ctx_w4s_min_cin input channels up, see apply_tuned)."""
    env = os.environ.get('CTDET_WINO_TILES')
    tiles = tuple(int(t) for t in (env or '2,4,44,46').split(',') if t)
    if env is None and getattr(backend, 'h2', False):
        tiles = tiles + H2_OF_TILE_VALUES
    allowed = getattr(backend, 'wino_tile_set', None)
    if allowed is not None:             # a runtime's accuracy policy: its set, narrowed by an explicit CTDET_WINO_TILES
        tiles = tuple(t for t in allowed if env is None or t in tiles)
        cap = getattr(backend, 'wino4_max_cin', None)
        f4 = ctx_f4_tile()
        if cap and st is not None and st.cin <= cap and f4 in WINO4F_TILES and not st.rt.get('wino4f_ok'):
            f4 = 4                      # the fused bf16x3 kernel needs 16-channel chunks: such layers keep the fp32 fused kernel
        if cap and st is not None and st.cin <= cap and f4 not in tiles and (env is None or str(f4) in env.split(',')):
            tiles = tiles + (f4,)       # short channel sums: F(4x4) costs little accuracy there
    if st is not None and not st.rt.get('winox_ok'):
        tiles = tuple(t for t in tiles if t not in WINOX_TILES)
    if st is not None and not st.rt.get('wino4s_ok'):
        tiles = tuple(t for t in tiles if t not in WINO4S_TILES)
    if st is not None and not st.rt.get('wino4f_ok'):
        tiles = tuple(t for t in tiles if t not in WINO4F_TILES)
    if st is not None and not st.rt.get('wino_ok'):          # dilated 3x3: only the three-kernel form
        tiles = tuple(t for t in tiles if t in WINO4S_TILES)
    return tiles


# bf16x3 tile -> the same kernel on the f16x2 operand form (csrc/ct_f16x2.h: two binary16 pieces, three products)
H2_OF_TILE = {44: 47, 46: 48}
H2_OF_TILE_VALUES = tuple(H2_OF_TILE.values())


H2_MIN_PIXELS = 8 * 300 * 300


def operand_form_h2(net, batch=None):
    """(winograd, direct): whether an inference runtime runs its bf16x3 Winograd table entries on the f16x2 operand form, and
    whether it may use the f16x2 twins of the direct kernel's tiles (csrc/ct_f16x2.h: two binary16 pieces, three products; same
    error against fp64 per layer, tests/test_gpu_wino.py::test_wino_rounding_error_vs_fp64, half the matrix instructions).
    CTDET_H2: '1' (default) = from batch x size^2 >= 8 x 300^2 up, '2' = always, '0' = never.  Why a threshold: an f16x2 launch
    waits for its input's maxima when it starts and folds its own in when it ends -- a few us per launch that the launch-bound
    small batches do not get back (same-box, images/s bf16x3 -> f16x2: RFBNet-300 bs 4 2 065 -> 1 830, bs 8 2 850 -> 2 990,
    bs 16 3 450 -> 3 945, bs 32 4 005 -> 4 655; RFBNet-512 bs 4 1 120 -> 1 126, bs 8 1 340 -> 1 438, bs 32 1 608 -> 1 805;
    profiles/r06_ab_batches.txt).  Networks with the Context-Transformer block (ctx_policy): under the shipped policy 'h2' the
    Winograd forms at EVERY batch size and never the direct twins -- the combination the parity sweeps were made on; under a
    tile-set policy neither."""
    mode = os.environ.get('CTDET_H2', '1')
    pol = ctx_policy(net)
    if mode == '0' or (pol is not None and pol not in ('h2', 'any')):
        return False, False
    if pol == 'h2':
        return True, False
    size = int(getattr(net, 'size', 300) or 300)
    on = mode == '2' or batch is None or batch * size * size >= H2_MIN_PIXELS
    return on, on and os.environ.get('CTDET_H2_X3', '1') != '0'


CTX_TILES_DEFAULT = 'h2'
SIDE_AFTER_DEFAULT = ''


def ctx_policy(net):
    """CTDET_CTX_TILES for a network with the Context-Transformer block (models/RFB_Net_vgg.py:253-271), None for any other:
    'h2' (default, round 6), 'any', or a comma list of Winograd tile codes (the round-2 .. 5 tile-set policies; '2,23' was
    round 5's)."""
    if not (getattr(net, 'method', None) == 'ours' and getattr(net, 'phase', 1) == 2):
        return None
    return os.environ.get('CTDET_CTX_TILES', CTX_TILES_DEFAULT)


def ctx_tile_set(net):
    """Winograd tile SET of networks with the Context-Transformer block; None = no restriction (every other network, and the
    policies 'h2' / 'any').

    The block's un-scaled theta.phi^T softmax is near-arg-max and amplifies a perturbation of its INPUT (the conf-head
    output) ~1000x (tools/ctx_parity.py --budget: 970x), so the reference's own fp32 CPU path sits 5..7e-5 from an fp64
    evaluation, two correct fp32 evaluations differ by up to ~1e-4, and which side of north_star's flat 1e-4 the worst of 7e5
    elements lands on is decided by single layers' summation orders: every policy is a MEASURED choice over the nine sweep cases
    (bs {2, 8, 32} x seeds {1234, 7, 99}) x the reference at 8 and 128 threads, not a guarantee for other seeds.

    Round 6, shipped: 'h2' -- the unconstrained table with its F(4x4,3x3) entries on the f16x2 operand form (three-kernel form with
    two accumulators, fused kernel) and the direct layers on bf16x3 with two accumulators (operand_form_h2).  All 18 pairs inside
    1e-4 (worst 9.76e-5), RFBNet-300 + Context-Transformer bs 32 at 3 640 images/s against 2 560 for round 5's policy; on ten
    further cases (seeds 1..5, bs 8 / 32) 2 of 20 pairs above 1e-4 against 5 of 20 for round 5's policy
    (profiles/r06_ctx_policy.txt, r06_ctx_policy_seeds.txt).  With the direct layers on f16x2 too: 3 750 images/s, one pair at
    1.01e-4.  Round 5, still available as CTDET_CTX_TILES=2,23: F(2x2,3x3) on bf16x3 with two accumulators (tile 23: per-layer
    error vs fp64 4e-7 against 2e-6 for F(4x4,3x3)) except a fused fp32 F(4x4) on the short channel sums (ctx_f4_max_cin): 18 / 18
    inside 1e-4 too (worst 9.2e-5), at 2 560 images/s.  Layers without 16-channel chunks keep F(2x2,3x3) on the fp32 MFMA."""
    v = ctx_policy(net)
    if v is None or v in ('any', 'h2'):
        return None
    return tuple(int(t) for t in v.split(',') if t)


CTX_F4_MAX_CIN_DEFAULT = '128'


def ctx_f4_max_cin(net):
    """Layers of a Context-Transformer network with at most this many input channels keep a fused F(4x4,3x3) kernel where the
    table picks one (ctx_f4_tile): its rounding error grows with the length of the channel sum, and on conv1_2 .. conv3_1
    (64 / 128 input channels at 300 x 300 .. 75 x 75) F(2x2,3x3) costs the most time.  Chosen by the round-5 sweeps
    (profiles/r05_ctx_policy.txt: RFBNet-300 + Context-Transformer bs 32, 9 randn cases, every case judged against the fp32 CPU
    path at 8 AND at 128 reference threads; three-kernel form off; (ctx_f4_tile, this cap) on the committed table):
      (4, 128)   2 570 images/s   worst GPU-CPU32 9.0e-5 at 8 threads, 9.2e-5 at 128   all 18 inside 1e-4   <- default
      (4, 256)   2 619            1.02e-4 / 1.07e-4    3 of 18 above 1e-4 (conv3_2 / conv3_3: 256-channel sums on the fp32 MFMA)
      (46, 128)  2 670            1.01e-4 / 9.5e-5     1 of 18 above
      (46, 256)  2 769            1.01e-4 / 9.7e-5     1 of 18 above
      cap 0 (every Winograd layer on F(2x2,3x3) / bf16x3): 2 353, 9.7e-5 / 1.02e-4, 1 of 18 above
      round 4's policy (three-kernel F(4x4) from 128 channels up): 2 898, 1.03e-4 / 1.05e-4, 3 of 18 above
    All of them are within 7.7e-5 of the fp64 evaluation; which side of 1e-4 the worst of 7e5 elements lands on against a
    reference that is itself 4.8..7.2e-5 from fp64 is decided by single layers' summation orders.  CTDET_CTX_F4_MAX_CIN; 0 = none."""
    return int(os.environ.get('CTDET_CTX_F4_MAX_CIN', CTX_F4_MAX_CIN_DEFAULT)) if ctx_tile_set(net) is not None else 0


CTX_F4_TILE_DEFAULT = '4'


def ctx_f4_tile():
    """Which F(4x4,3x3) kernel the layers below ctx_f4_max_cin run: 4 = fused on the fp32 MFMA (csrc/ct_wino4.hip), 46 = fused on
    bf16x3 (csrc/ct_wino4f.hip; layers without 16-channel chunks keep 4).  CTDET_CTX_F4_TILE."""
    return int(os.environ.get('CTDET_CTX_F4_TILE', CTX_F4_TILE_DEFAULT))


CTX_W4S_MIN_CIN_DEFAULT = '0'
# Context-Transformer networks with ctx_w4s_min_cin > 0: their dilated layers (conv6, the RFB branches) on the three-kernel form
# too, from that many input channels up (CTDET_CTX_DIL_W4S=0: never)
CTX_DIL_W4S_DEFAULT = '1'


def ctx_w4s_min_cin(net):
    """Layers of a Context-Transformer network with at least this many input channels that the table runs on F(4x4,3x3)
    (fused or three-kernel) use the three-kernel bf16x3 form with two accumulators (tile 44: error vs fp64 2e-6 per layer
    against 3-4e-7 for F(2x2,3x3) / bf16x3 with two accumulators, at 1.7x the speed on the wide layers).  Default 0 = never:
    with tile 44 on the 512-channel layers one to three of the 18 (case, reference thread count) pairs of the sweep land at
    1.03-1.05e-4 from the fp32 CPU path (ctx_f4_max_cin has the table), and north_star's contract is a flat 1e-4.
    CTDET_CTX_W4S_MIN_CIN=128 CTDET_CTX_F4_MAX_CIN=128 is the round-4 policy: +17 % images/s for callers who accept that."""
    return int(os.environ.get('CTDET_CTX_W4S_MIN_CIN', CTX_W4S_MIN_CIN_DEFAULT)) if ctx_tile_set(net) is not None else 0


def apply_tuned(backend, st, batch, wino4=True):
    """Give a prepared conv step the committed tile choice for its shape; False if the table has none.
    wino4=False maps a 'wino4' entry to 'wino'.  A Winograd entry the runtime's policy excludes (ctx_tile_set) becomes the
    most accurate allowed variant: F(2x2) on bf16x3 with two accumulators where the layer has 16-channel chunks, else
    F(2x2) on the fp32 MFMA."""
    cfg = tune_table().get(st.tune_key(batch))
    if getattr(backend, 'h2', False):
        # a runtime on the f16x2 operand forms: where the forms' different speed-ups change which KERNEL FAMILY wins a shape
        # (tools/tune_convs.py --h2), the table holds that choice under '<key>|h2' (same names: mapped to the f16x2 twins below)
        cfg = tune_table().get(st.tune_key(batch) + '|h2', cfg)
    names = [backend.lib.ct_conv_config_name(i).decode() for i in range(backend.lib.ct_conv_num_configs())]
    codes = {v: k for k, v in WINO_NAME.items()}
    usable = cfg in codes and (st.rt.get('wino_ok') or (codes[cfg] in WINO4S_TILES and st.rt.get('wino4s_ok'))) and \
        os.environ.get('CTDET_WINO', '1') != '0'
    if usable and st.dil > 1:
        # dilated layer on the three-kernel form: where tile 44 is allowed as such; a runtime with an accuracy policy
        # (Context-Transformer networks) takes it from ctx_w4s_min_cin input channels up (CTDET_CTX_DIL_W4S=0: never --
        # the layer then runs the table's previous choice, '|alt').  Only tile 44 (and its f16x2 twin 47) exists for these layers: a caller
        # that rules out F(4x4) (wino4=False) gets the '|alt' entry as well.
        policy = getattr(backend, 'wino_tile_set', None) is not None
        if policy:
            usable = os.environ.get('CTDET_CTX_DIL_W4S', CTX_DIL_W4S_DEFAULT) != '0' and st.cin >= getattr(backend, 'ctx_w4s_min_cin', 0) > 0
        else:
            usable = codes[cfg] in wino_tiles(backend, st)
        usable = usable and wino4
    if cfg in codes and not usable:
        cfg = tune_table().get(st.tune_key(batch) + '|alt')       # what the layer ran on before the three-kernel form took it
    if usable:
        allowed = wino_tiles(backend, st)
        want = codes[cfg]
        if want in F4_TILES and not wino4:
            want = 2
        if getattr(backend, 'wino_tile_set', None) is not None:
            # accuracy policy of this runtime: F(4x4) / fp32 survives only where the policy allows it (short channel sums),
            # everything else runs the most accurate allowed variant
            w4s_min = getattr(backend, 'ctx_w4s_min_cin', 0)
            f4 = ctx_f4_tile()
            f4 = f4 if f4 in allowed else 4     # tile 46 needs 16-channel chunks
            if want in F4_TILES and w4s_min and st.cin >= w4s_min and st.rt.get('wino4s_ok'):
                want = 44                   # three-kernel F(4x4) with two accumulators: 0.4x the rounding of the fused fp32 form
            elif want in (4,) + WINO4F_TILES and f4 in allowed:
                want = f4                   # a fused F(4x4) entry below ctx_f4_max_cin input channels (wino_tiles put f4 into the set)
            else:
                want = 23 if 23 in allowed else 2 if 2 in allowed or not allowed else allowed[0]
        elif want in WINO4F_TILES and st.cin > (getattr(backend, 'w4f_max_cin', None) or 1 << 30):
            # the training runtime of a Context-Transformer network (TrainRuntime sets w4f_max_cin = 128): the fused bf16x3
            # kernel has ONE accumulator, 3.4e-6 of the output range at 256 input channels and 4.6e-6 at 512 against 1.3-2.0e-6
            # for the three-kernel form, and the block's backward amplifies that (conf.3's gradient 2.5e-4 from fp64 instead of
            # <= 1.4e-4 at RFBNet-512 bs 8) -- its wide layers stay on the three-kernel form
            want = 44 if st.rt.get('wino4s_ok') and 44 in allowed else 4 if 4 in allowed else 2
        elif want not in allowed:
            # a three-kernel / fused-bf16x3 F(4x4) entry without its tile in the set (CTDET_WINO_TILES=2,4) is the fused fp32 F(4x4) kernel's layer
            want = 4 if want in WINO4S_TILES + WINO4F_TILES and 4 in allowed else 2 if 2 in allowed or not allowed else allowed[0]
        if getattr(backend, 'h2', False) and H2_OF_TILE.get(want) in allowed:
            want = H2_OF_TILE[want]         # the same kernel on the f16x2 operand form (operand_form_h2)
        backend.enable_wino(st, tile=want)
        return True
    if isinstance(cfg, str) and cfg.startswith('h2:'):
        # the f16x2 twin of a direct-kernel tile: only from a '<key>|h2' entry (tools/tune_convs.py --h2 times it against the
        # bf16x3 tile per shape: it wins from batch 8-16 up, not on the launch-bound small batches)
        xn = backend.x3_names()
        if getattr(backend, 'h2_direct', False) and cfg in xn and x3_allowed(st) and st.cin % backend.x3_bk(xn.index(cfg)) == 0:
            backend.enable_x3(st, xn.index(cfg))
            return True
        cfg = 'x3:' + cfg[3:]
    if isinstance(cfg, str) and cfg.startswith('x3:'):
        xn = backend.x3_names()
        if cfg in xn and x3_allowed(st) and st.cin % backend.x3_bk(xn.index(cfg)) == 0:     # the k-step must divide cin
            use = cfg
            backend.enable_x3(st, xn.index(use))
            return True
        cfg = tune_table().get(st.tune_key(batch) + '|f32')       # the best fp32-MFMA tile, recorded next to it
    if cfg == 'valu' and not (st.cin == 3 and (st.kh, st.kw, st.stride, st.dil) == (3, 3, 1, 1) and st.res is None):
        cfg = None                      # the vector-ALU kernel exists for the 3-channel image layer only
    if cfg in names:
        st.rt['config'] = names.index(cfg) + 1
        st.rt['desc'].config = st.rt['config']
        return True
    return False


def run_on_streams(rt, run_step):
    """Walk the plan's steps in launch order (rt.order: plan order, side-stream steps possibly deferred, see
    Runtime._build_schedule), each step on the stream the schedule gave it (rt.sid), cross-stream producer -> consumer
    edges as events (rt.xdeps / rt.signal / rt.ev); everything joins the caller's stream."""
    steps = rt.plan.steps
    if rt.side is None:
        for st in steps:
            run_step(st)
        return
    main = torch.cuda.current_stream(rt.backend.device)
    for sd in rt.sides:
        sd.wait_stream(main)
    streams = [main] + rt.sides
    order = getattr(rt, 'order', None) or list(range(len(steps)))
    pos, n = 0, len(order)
    while pos < n:
        k = rt.sid[order[pos]]
        with torch.cuda.stream(streams[k]):
            while pos < n and rt.sid[order[pos]] == k:
                i = order[pos]
                for j in rt.xdeps[i]:
                    streams[k].wait_event(rt.ev[j])
                run_step(steps[i])
                if i in rt.signal:
                    rt.ev[i].record(streams[k])
                pos += 1
    for sd in rt.sides:
        main.wait_stream(sd)


class Runtime:
    """A Plan bound to a backend: buffers, packed weights, and the forward entry points."""

    def __init__(self, net, batch, backend, tune=None):
        self.plan = Plan(net, batch)
        self.backend = backend
        self.batch = batch
        self.net = net
        bufs = {}
        for name, shp in self.plan.buf_shapes.items():
            bufs[name] = backend.alloc((batch,) + tuple(shp))
        self.bufs = bufs
        for st in self.plan.steps:
            if st.kind == 'conv':
                backend.prepare_conv(st, bufs, batch)
        # tile config per conv: committed table first (names, so it survives config reordering),
        # live autotune only for shapes the table does not know (CTDET_TUNE=0 disables, =2 forces)
        mode = os.environ.get('CTDET_TUNE', '1') if tune is None else ('1' if tune else '0')
        backend.wino_tile_set = ctx_tile_set(net)
        backend.h2, backend.h2_direct = operand_form_h2(net, batch)
        if os.environ.get('CTDET_CTX_W4F_MAX_CIN') and getattr(net, 'method', None) == 'ours' and getattr(net, 'phase', 1) == 2:
            backend.w4f_max_cin = int(os.environ['CTDET_CTX_W4F_MAX_CIN'])      # experiments: fused one-accumulator kernel only up to here
        backend.wino4_max_cin = ctx_f4_max_cin(net)
        backend.ctx_w4s_min_cin = ctx_w4s_min_cin(net)
        self.tuned = False
        self.live_tuned = []
        self.event_log = None        # set to a list to collect (step, start_event, end_event) per conv
        if getattr(backend, 'tune_conv', None) is not None:
            missing = [st for st in self.conv_steps() if mode == '2' or not apply_tuned(backend, st, batch)]
            if missing and mode != '0':
                self.autotune(missing)
            self.tuned = not missing or mode != '0'
            # shapes the committed table does not serve: timed live (mode 1 / 2) or left to the library's heuristic (mode 0) --
            # either way a tile choice no parity test pinned at that shape (bench.py refuses a headline with any)
            self.live_tuned = [st.tune_key(batch) for st in missing]
            # CTDET_WINO_FORCE = tile code (experiments, tools/ctx_parity.py): every layer that runs on a Winograd kernel
            # and has the geometry for it is moved to that variant
            force = int(os.environ.get('CTDET_WINO_FORCE', '0') or 0)
            for st in (self.conv_steps() if force else ()):
                # dilated layers only exist on the three-kernel form: they keep it unless that is what is being forced
                geo = st.rt.get('wino4s_ok') if force in WINO4S_TILES else st.rt.get('wino4f_ok') if force in WINO4F_TILES else \
                    st.rt.get('wino_ok')
                if st.rt.get('wino') and geo and (st.rt.get('winox_ok') or force in (2, 4) + WINO4S_TILES + WINO4F_TILES):
                    backend.enable_wino(st, tile=force)
        self._fuse_pools()
        self._mark_exclusive()
        self._build_schedule()
        self._share_workspaces()
        self._wire_absmax()

    def _fuse_pools(self):
        """MaxPool2d(2, 2) directly behind a Winograd conv: the 2x2 output tile is the pooling window, so
        the conv writes the pooled map itself (and skips its full-resolution output if nothing else reads it)."""
        steps = self.plan.steps
        fuse = os.environ.get('CTDET_FUSE_POOL', '1') != '0'
        for pi, ps in enumerate(steps):
            if ps.kind != 'pool' or (ps.k, ps.stride, ps.pad) != (2, 2, 0):
                continue
            prod = [st for st in steps if st.kind == 'conv' and not st.segs and st.dst == ps.src]
            if len(prod) != 1 or not prod[0].rt.get('wino') or prod[0].dst_coff != 0 or prod[0].cout != ps.ch \
                    or prod[0].res is not None:
                continue
            # a fusable producer never splits its channel sum (the fused form cannot): CTDET_FUSE_POOL=0 then
            # changes the launches, not the arithmetic
            prod[0].rt['desc'].ksplit = 0
            if not fuse:
                continue
            others = [st for st in steps if st is not ps and (getattr(st, 'src', None) == ps.src or
                                                              getattr(st, 'res', None) == ps.src)]
            prod[0].rt['pool'] = (self.bufs[ps.dst], ps.oh, ps.ow, bool(others))
            ps.fused_into = prod[0].name

    def _mark_exclusive(self):
        """Opt-in (CTDET_W4_STREAMK=1): F(4x4,3x3) launches of the trunk up to the first branch point (conv1_1 ..
        conv4_3) have the device to themselves in every schedule -- nothing the side stream runs exists before the
        first Norm / head source -- so they may use the kernel's stream-K form (desc.ksplit = -2: a persistent grid
        whose last, partial round is cut by input-channel chunks; 67 MB of slab workspace shared by those launches,
        they run one after another).  +1.3 .. 2.5 % images/s at bs 32 (three layers with 3.1 rounds of workgroups).
        Off by default: the cut items are the LAST tiles of the batch, so an image's last bits would depend on its
        position in the batch (tests/test_gpu_harness.py and test_full_size_pipeline_properties check that they do not).
        The marking does not depend on CTDET_STREAMS or CTDET_FUSE_POOL."""
        if os.environ.get('CTDET_W4_STREAMK', '0') != '1' or not hasattr(self.backend, 'lib'):
            return
        steps = self.plan.steps
        first_side = next((i for i, st in enumerate(steps)
                           if st.name.startswith(('Norm.', 'head.')) or st.kind == 'ctxpool'), len(steps))
        ws = None
        for st in steps[:first_side]:
            if st.kind != 'conv' or st.rt.get('wino') != 4:
                continue
            if ws is None:
                ws = self.bufs.setdefault('__streamk_ws', self.backend.alloc((256 * 2 * 64 * 32 * 16,)))
            d = st.rt['desc']
            d.ksplit, d.ksplit_ws, d.ksplit_ws_floats = -2, ws.data_ptr(), ws.numel()

    # ---- two-stream schedule: the Norm branch and the multibox heads are independent of the trunk that
    # follows their source (base.23.., extras..), and the small 19x19 .. 1x1 kernels of that trunk cannot fill
    # 256 CUs on their own; running the side work on a second HIP stream lets it soak up the idle CUs.
    def _build_schedule(self):
        self.side = None
        steps = self.plan.steps
        if os.environ.get('CTDET_STREAMS', '2') == '1' or self.backend.device.type != 'cuda':
            return
        nstreams = int(os.environ.get('CTDET_STREAMS', '2'))

        def stream_of(st):
            side = st.name.startswith(('Norm.', 'head.')) or st.kind == 'ctxpool'
            sid = 1 if side else 0
            if nstreams > 2:                     # the independent branches of an RFB block on their own streams
                for b, tag in enumerate(('.b1.', '.b2.', '.b3.')):
                    if tag in st.name:
                        sid = 2 + (3 if side else 0) + b
            return sid
        sid = [stream_of(st) for st in steps]
        if not any(sid):
            return

        def reads(st):
            r = [st.src]
            if getattr(st, 'res', None) is not None:
                r.append(st.res)
            return r

        def writes(st):
            if st.kind == 'conv' and st.segs:
                return [sg.dst for sg in st.segs]
            if st.kind == 'conv' and st.rt.get('pool') is not None:
                return [st.dst] + [p.dst for p in steps if getattr(p, 'fused_into', None) == st.name]
            return [st.dst]
        writers = {}
        self.xdeps, self.signal = [[] for _ in steps], set()
        for i, st in enumerate(steps):
            for b in reads(st):
                for j in writers.get(b, []):
                    if sid[j] != sid[i]:
                        self.xdeps[i].append(j)
                        self.signal.add(j)
            for b in writes(st):
                writers.setdefault(b, []).append(i)
        self.sid = sid
        # Launch order.  The side work (Norm branch, heads: 38x38 / 19x19 layers that fill the chip) only depends on
        # conv4_3 / conv7, while the END of the trunk (extras.1 .. extras.6 and their heads: 10x10 .. 1x1 maps, ~30 launches
        # of 25 .. 80 workgroups) cannot fill 256 CUs.  CTDET_SIDE_AFTER = name of a trunk step: the side-stream steps that
        # precede it in plan order are launched right after it instead, so they run beside the small layers rather than
        # beside conv5 .. conv7, which fill the chip on their own.
        self.order = list(range(len(steps)))
        after = os.environ.get('CTDET_SIDE_AFTER', SIDE_AFTER_DEFAULT)
        names = [st.name for st in steps]
        if after and after in names:
            k = names.index(after)
            # only the side stream proper (Norm / heads / context pooling) is deferred: with CTDET_STREAMS > 2 the RFB branch
            # streams feed main-stream steps at or before the anchor, whose waits must see this pass's events
            early = [i for i in range(k) if sid[i] == 1]
            self.order = [i for i in range(k + 1) if sid[i] != 1] + early + list(range(k + 1, len(steps)))
        self.sides = [torch.cuda.Stream(self.backend.device) for _ in range(max(sid))]
        self.side = self.sides[0]
        self.ev = {j: torch.cuda.Event() for j in self.signal}

    def _wire_absmax(self):
        """Per-image maxima of |activation| for the f16x2 kernels (ct_conv_desc.in_absmax / out_absmax, csrc/ct_f16x2.h).  A buffer
        that an f16x2 layer reads gets a slot (one line per image) when EVERY step that writes it folds the maxima of what it
        stores into one: every forward convolution kernel does, except the fused Winograd kernels on fp32 / bf16x3; a max-pool's
        output shares the slot of its input (pooled values are a subset), whether the pool runs as a step or inside its producer.
        An f16x2 consumer whose buffer has no slot runs the bf16x3 twin of its kernel (tile 47 takes the maxima itself, inside its
        launch).  The slots are zeroed once per step (run_loaded)."""
        be = self.backend
        if not hasattr(be, 'new_slot'):
            return
        steps = self.plan.steps

        def consumes(st):
            return st.rt.get('wino') in H2_TILES or (not st.rt.get('wino') and st.rt.get('x3') is not None and be.x3_h2(st.rt['x3']))

        def tracks(st):
            if st.kind != 'conv' or st.segs:
                return False
            return st.rt.get('wino') in TRACK_TILES if st.rt.get('wino') else True
        root = {}                          # pooled buffer -> the buffer whose maximum bounds it
        for ps in steps:
            if ps.kind == 'pool':
                root[ps.dst] = ps.src

        def root_of(b):
            while b in root:
                b = root[b]
            return b
        writers = {}
        for st in steps:
            if st.kind == 'conv':
                for b in ([sg.dst for sg in st.segs] if st.segs else [st.dst]):
                    writers.setdefault(b, []).append(st)
            elif st.kind != 'pool':
                writers.setdefault(st.dst, []).append(st)
        while True:
            for st in self.conv_steps():
                st.rt['desc'].in_absmax = None
                st.rt['desc'].out_absmax = None
                st.rt.pop('amax_own', None)
                st.rt.pop('amax_frozen', None)
            slots, fallback = {}, []
            be.slots_used = 0
            for st in self.conv_steps():
                if not consumes(st):
                    continue
                b = root_of(st.src)
                ws = writers.get(b, [])
                if ws and all(tracks(w) for w in ws):
                    if b not in slots:
                        slots[b] = be.new_slot(self.batch)
                        for w in ws:
                            w.rt['desc'].out_absmax = slots[b]
                    st.rt['desc'].in_absmax = slots[b]
                elif st.rt.get('wino') != 47:
                    fallback.append(st)
            if not fallback:
                break
            for st in fallback:            # (a layer that leaves the f16x2 form may stop tracking: wire again)
                if st.rt.get('wino') == 48:
                    be.enable_wino(st, tile=46)
                else:
                    xn = be.x3_names()
                    be.enable_x3(st, xn.index('x3:' + xn[st.rt['x3']][3:]))
        self._wired_epoch = be.kernel_epoch
        self.amax_slots = slots

    def _share_workspaces(self):
        """One three-kernel-Winograd workspace per stream of the schedule (HipBackend.ws_pool) instead of one per layer."""
        if not hasattr(self.backend, 'ws_rebuild'):
            return
        for i, st in enumerate(self.plan.steps):
            if st.kind == 'conv':
                st.rt['ws_key'] = self.sid[i] if self.side is not None else 0
        self.backend.ws_rebuild(self.conv_steps())

    def autotune(self, steps=None):
        self.bufs['x'].normal_()
        for st in self.plan.steps:            # run once so every buffer holds realistic data
            self._run_step(st)
        for st in (steps if steps is not None else self.conv_steps()):
            self.backend.tune_conv(st)

    def tuned_configs(self):
        lib = self.backend.lib
        out = {}
        xn = self.backend.x3_names()
        for st in self.conv_steps():
            if not (st.rt['desc'].config > 0 or st.rt.get('wino') or st.rt.get('x3') is not None):
                continue
            key = st.tune_key(self.batch)
            f32 = lib.ct_conv_config_name(st.rt['desc'].config - 1).decode() if st.rt['desc'].config > 0 else None
            if st.rt.get('wino'):
                out[key] = WINO_NAME[st.rt['wino']]
                if st.dil > 1 and st.rt.get('alt'):       # apply_tuned's fallback where tile 44 may not be used on a dilated layer
                    out[key + '|alt'] = st.rt['alt']
                    if f32 and st.rt['alt'].startswith('x3:'):
                        out[key + '|f32'] = f32
            elif st.rt.get('x3') is not None:
                out[key] = xn[st.rt['x3']]
                if f32:
                    out[key + '|f32'] = f32        # fallback tile when CTDET_X3=0
            else:
                out[key] = f32
        return out

    def refresh_weights(self):
        """Re-pack any fused conv whose parameters changed since the last pack."""
        for st in self.plan.steps:
            if st.kind == 'conv' and st.rt.get('versions') != self.backend.param_versions(st):
                self.backend.pack_conv(st)

    def _run_step(self, st):
        if st.kind == 'conv':
            if self.event_log is not None:      # HIP events on the launch stream (bench roofline)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self.backend.run_conv(st)
                e1.record()
                self.event_log.append((st, e0, e1))
                return
            self.backend.run_conv(st)
        elif st.kind == 'pool':
            if getattr(st, 'fused_into', None) is None:
                self.backend.run_pool(st, self.bufs, self.batch)
        elif st.kind == 'ctxpool':
            self.backend.run_ctxpool(st, self.bufs, self.batch)
        else:
            raise ValueError(st.kind)

    def load_input(self, x):
        """Shape check, re-pack of changed weights, copy of x [B,3,S,S] into the plan's input buffer."""
        want = (self.batch,) + tuple(self.plan.buf_shapes['x'])
        if tuple(x.shape) != want:
            raise _lib.CtdetError('plan was built for input %s, got %s' % (want, tuple(x.shape)))
        self.refresh_weights()
        if hasattr(self.backend, 'load_input'):       # bf16 path: NCHW fp32 image -> NHWC bf16
            self.backend.load_input(self.bufs['x'], x)
        else:
            self.bufs['x'].copy_(x)

    def run_loaded(self):
        """The launches of the backbone on the input buffer (no allocation, no host synchronisation: capturable)."""
        if getattr(self.backend, 'zero_slots', None) is not None:
            self.ensure_wired()
            self.backend.zero_slots()       # the maxima of |activation| the f16x2 layers take their scales from (_wire_absmax)
        run_on_streams(self, self._run_step)
        return self.bufs['loc'], self.bufs['conf'], self.bufs['obj']

    def ensure_wired(self):
        """Re-wire the maxima slots when a step changed kernels since they were wired (tuner, tools): a producer may have stopped
        tracking.  Allocates: a DetectionPipeline calls it before it captures or replays a hipGraph."""
        if getattr(self.backend, 'new_slot', None) is not None and getattr(self, '_wired_epoch', None) != self.backend.kernel_epoch:
            self._wire_absmax()

    def run_backbone(self, x):
        """x [B,3,S,S] on the device -> raw (loc [B,P*4], conf [B,P*C], obj [B,P*2]) buffers (views)."""
        self.load_input(x)
        return self.run_loaded()

    def policy_record(self):
        """Every switch that shaped this runtime's kernel choice, resolved to what is in force (bench.py prints it as
        `config.policy`; a line measured under a non-default switch says so).  `env` lists the CTDET_* variables that are set --
        the library reads a few of its own (launch geometry experiments), so they are named even where Python ignores them."""
        b = self.backend
        tiles = {}
        for st in self.conv_steps():
            w = st.rt.get('wino')
            if w:
                tiles[int(w)] = tiles.get(int(w), 0) + 1
        return {
            'operand_form': 'f16x2' if getattr(b, 'h2', False) else 'bf16x3',
            'direct_twins_f16x2': bool(getattr(b, 'h2_direct', False)),
            'ctx_tiles': ctx_policy(self.net),
            'wino_tile_set': sorted(b.wino_tile_set) if getattr(b, 'wino_tile_set', None) is not None else None,
            'winograd_layers_by_tile': {str(k): v for k, v in sorted(tiles.items())},
            'tune_table': os.path.basename(TUNE_TABLE),
            'live_tuned_layers': len(self.live_tuned),
            'env': {k: v for k, v in sorted(os.environ.items()) if k.startswith('CTDET_')},
        }

    def conv_steps(self):
        return [s for s in self.plan.steps if s.kind == 'conv']
