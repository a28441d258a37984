"""Tensor-level wrappers over the libctdet C ABI.

torch is plumbing here (device memory + the current HIP stream); every function launches
hand-written HIP kernels through ctypes and raises if the tensors are not on a HIP device --
there is deliberately NO CPU / PyTorch fallback for these ops.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, lib


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.CtdetError('%s must be a tensor on the HIP device (got %s); the ctdet ops have no '
                              'CPU fallback' % (name, getattr(t, 'device', type(t))))
    if t.dtype != dtype:
        raise _lib.CtdetError('%s must be %s, got %s' % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise _lib.CtdetError('%s must be contiguous' % name)
    return C.c_void_p(t.data_ptr())


def _opt(t, name, dtype=torch.float32):
    return C.c_void_p(0) if t is None else _dev(t, name, dtype)


# ------------------------------------------------------------------ boxes
def decode(loc, priors, variances, scale=None):
    """utils/box_utils.py:184-202 on [P,4] or [B,P,4]; optional `boxes *= scale` ([4] or [B,4])."""
    batched = loc.dim() == 3
    B = loc.shape[0] if batched else 1
    P = priors.shape[0]
    out = torch.empty_like(loc)
    per_image = int(scale is not None and scale.dim() == 2)
    check(lib().ct_decode(_dev(loc, 'loc'), _dev(priors, 'priors'), B, P, float(variances[0]),
                          float(variances[1]), _opt(scale, 'scale'), per_image, _dev(out, 'out'), _stream()),
          'ct_decode')
    return out


def encode(matched, priors, variances):
    out = torch.empty_like(matched)
    check(lib().ct_encode(_dev(matched, 'matched'), _dev(priors, 'priors'), priors.shape[0],
                          float(variances[0]), float(variances[1]), _dev(out, 'out'), _stream()), 'ct_encode')
    return out


def detect_fused(loc, conf, obj, priors, variances, apply_softmax=False, scale=None, out=None):
    """layers/functions/detection.py:18-55 (+ eval softmaxes when apply_softmax, + the
    `boxes *= scale` of test.py:136 when scale ([4] or [B,4]) is given)."""
    B, P = loc.shape[0], priors.shape[0]
    Cn = conf.shape[-1]
    if out is None:
        boxes = torch.empty(B, P, 4, device=loc.device, dtype=torch.float32)
        scores = torch.empty(B, P, Cn + 1, device=loc.device, dtype=torch.float32)
    else:
        boxes, scores = out
    per_image = int(scale is not None and scale.dim() == 2)
    check(lib().ct_detect_fused(_dev(loc, 'loc'), _dev(conf, 'conf'), _dev(obj, 'obj'), _dev(priors, 'priors'),
                                B, P, Cn, float(variances[0]), float(variances[1]), int(apply_softmax),
                                _opt(scale, 'scale'), per_image,
                                _dev(boxes, 'boxes'), _dev(scores, 'scores'), _stream()), 'ct_detect_fused')
    return boxes, scores


def softmax_lastdim(x):
    out = torch.empty_like(x)
    cols = x.shape[-1]
    check(lib().ct_softmax_lastdim(_dev(x, 'x'), _dev(out, 'out'), x.numel() // cols, cols, _stream()),
          'ct_softmax_lastdim')
    return out


def jaccard(box_a, box_b, b_center_form=False):
    out = torch.empty(box_a.shape[0], box_b.shape[0], device=box_a.device, dtype=torch.float32)
    if out.numel() == 0:
        return out
    check(lib().ct_jaccard(_dev(box_a, 'box_a'), box_a.shape[0], _dev(box_b, 'box_b'), box_b.shape[0],
                           int(b_center_form), _dev(out, 'out'), _stream()), 'ct_jaccard')
    return out


def match_batched(targets, priors, threshold, variances, want_overlap=False):
    """targets: list of [G,6] tensors (x1,y1,x2,y2,label,weight) -> loc_t, conf_t, obj_t[, overlap]."""
    dev = priors.device
    B, P = len(targets), priors.shape[0]
    counts = [int(t.shape[0]) for t in targets]
    max_gt = max(max(counts), 1)
    off = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32).to(dev)
    if sum(counts) > 0:
        truths = torch.cat([t.reshape(-1, 6) for t in targets], 0).to(dev, torch.float32).contiguous()
    else:
        truths = torch.zeros(1, 6, device=dev)
    loc_t = torch.empty(B, P, 4, device=dev)
    conf_t = torch.empty(B, P, 2, device=dev)
    obj_t = torch.empty(B, P, device=dev, dtype=torch.uint8)
    overlap = torch.empty(B, P, device=dev) if want_overlap else None
    ws_bytes = lib().ct_match_workspace_bytes(B, P, max_gt)
    ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
    check(lib().ct_match_batched(_dev(truths, 'truths'), _dev(off, 'gt_off', torch.int32), B, max_gt,
                                 _dev(priors, 'priors'), P, float(threshold), float(variances[0]),
                                 float(variances[1]), _dev(loc_t, 'loc_t'), _dev(conf_t, 'conf_t'),
                                 _dev(obj_t, 'obj_t', torch.uint8), _opt(overlap, 'overlap'),
                                 _dev(ws, 'ws', torch.uint8), ws_bytes, _stream()), 'ct_match_batched')
    return (loc_t, conf_t, obj_t.bool(), overlap) if want_overlap else (loc_t, conf_t, obj_t.bool())


# ------------------------------------------------------------------ NMS
def nms_sorted_host(dets_sorted, thresh, ge=False, device_id=0, plain_iou=False):
    """The `_nms` contract (utils/nms/nms_kernel.cu:91-144): host numpy [n,dim>=4] sorted by
    descending score -> ascending kept indices (int32)."""
    d = np.ascontiguousarray(dets_sorted, dtype=np.float32)
    n = d.shape[0]
    keep = np.empty(max(n, 1), dtype=np.int32)
    num = C.c_int(0)
    check(lib().ct_nms_sorted_host_mode(keep.ctypes.data_as(C.c_void_p), C.cast(C.byref(num), C.c_void_p),
                                        d.ctypes.data_as(C.c_void_p), n, d.shape[1] if n else 5,
                                        float(thresh), int(bool(ge)) | (2 if plain_iou else 0),
                                        int(device_id)), 'ct_nms_sorted_host')
    return keep[:num.value]


def nms_batched(dets, seg_off, thresh, ge=False, max_seg_len=None):
    """dets dev [total,5] (segments sorted by score), seg_off dev int32 [S+1] -> keep [total], count [S]."""
    S = seg_off.numel() - 1
    keep = torch.empty(max(dets.shape[0], 1), device=dets.device, dtype=torch.int32)
    cnt = torch.empty(S, device=dets.device, dtype=torch.int32)
    check(lib().ct_nms_batched_dev(_dev(dets, 'dets'), _dev(seg_off, 'seg_off', torch.int32), S,
                                   int(max_seg_len or dets.shape[0]), float(thresh), int(bool(ge)),
                                   _dev(keep, 'keep', torch.int32), _dev(cnt, 'cnt', torch.int32),
                                   C.c_void_p(0), 0, _stream()), 'ct_nms_batched_dev')
    return keep, cnt


def cpu_nms(dets, thresh, ge=True):
    """utils/nms/cpu_nms.pyx:17-68 on a host numpy array (the reference's --cpu path)."""
    d = np.ascontiguousarray(dets, dtype=np.float32)
    n = d.shape[0]
    keep = np.empty(max(n, 1), dtype=np.int32)
    num = C.c_int(0)
    check(lib().ct_cpu_nms(d.ctypes.data_as(C.c_void_p), n, float(thresh), int(bool(ge)),
                           keep.ctypes.data_as(C.c_void_p), C.cast(C.byref(num), C.c_void_p)), 'ct_cpu_nms')
    return keep[:num.value]


def cpu_soft_nms(boxes, sigma=0.5, Nt=0.3, threshold=0.001, method=0):
    """utils/nms/cpu_nms.pyx:70-163: mutates `boxes` (float32 [n,5], C-contiguous) in place."""
    if not (isinstance(boxes, np.ndarray) and boxes.dtype == np.float32 and boxes.flags['C_CONTIGUOUS']):
        raise ValueError('cpu_soft_nms needs a C-contiguous float32 array (it is updated in place)')
    n_out = C.c_int(0)
    check(lib().ct_cpu_soft_nms(boxes.ctypes.data_as(C.c_void_p), boxes.shape[0], float(sigma), float(Nt),
                                float(threshold), int(method), C.cast(C.byref(n_out), C.c_void_p)),
          'ct_cpu_soft_nms')
    return n_out.value


class PostProcessor:
    """Batched test.py:136-161 on the device with persistent buffers."""

    def __init__(self, batch, num_priors, num_fg, device, out_cap=None):
        self.B, self.P, self.T = batch, num_priors, num_fg
        self.cap = int(out_cap or min(num_priors, 2048))
        self.ws_bytes = lib().ct_postprocess_workspace_bytes(batch, num_priors, num_fg)
        self.ws = torch.empty(self.ws_bytes, device=device, dtype=torch.uint8)
        self.out_dets = torch.zeros(batch, num_fg, self.cap, 5, device=device)
        self.out_count = torch.zeros(batch, num_fg, device=device, dtype=torch.int32)
        self.out_index = torch.zeros(batch, num_fg, self.cap, device=device, dtype=torch.int32)
        self.overflow = torch.zeros(1, device=device, dtype=torch.int32)

    def run(self, boxes, scores, conf_thresh=0.01, nms_thresh=0.45, ge=False, max_per_image=200):
        check(lib().ct_postprocess_batched(
            _dev(boxes, 'boxes'), _dev(scores, 'scores'), self.B, self.P, self.T, float(conf_thresh),
            float(nms_thresh), int(bool(ge)), int(max_per_image), self.cap, _dev(self.out_dets, 'out_dets'),
            _dev(self.out_count, 'out_count', torch.int32), _dev(self.out_index, 'out_index', torch.int32),
            _dev(self.overflow, 'overflow', torch.int32), _dev(self.ws, 'ws', torch.uint8), self.ws_bytes,
            _stream()), 'ct_postprocess_batched')
        return self.out_dets, self.out_count

    def to_all_boxes(self):
        """Host view in the reference's layout: all_boxes[img][cls] = float32 [k,5] (cls 0 empty)."""
        if int(self.overflow.item()):
            raise _lib.CtdetError('postprocess output capacity %d exceeded; raise out_cap' % self.cap)
        cnt = self.out_count.cpu().numpy()
        dets = self.out_dets.cpu().numpy()
        res = []
        for b in range(self.B):
            per = [np.empty((0, 5), dtype=np.float32)]
            for c in range(self.T):
                per.append(dets[b, c, :cnt[b, c]].copy())
            res.append(per)
        return res


# ------------------------------------------------------------------ input transform
class Preprocessor:
    """Batched BaseTransform (data/data_augment.py:224-266) on the device: uint8 HxWx3 images of
    any size -> float32 [B,3,S,S] (bilinear resize, minus means, CHW) in one launch.
    The packed bytes go through one pinned staging buffer and one H2D copy per batch."""

    def __init__(self, size, means, device, max_batch=32, max_pixels=512 * 512):
        self.size, self.device, self.max_batch = size, torch.device(device), max_batch
        self.means = (C.c_float * 3)(*[float(m) for m in means])
        self.cap = max_batch * max_pixels * 3
        self.stage = torch.empty(self.cap, dtype=torch.uint8).pin_memory()
        self.dev = torch.empty(self.cap, dtype=torch.uint8, device=self.device)
        self.meta_h = torch.empty(max_batch * 4, dtype=torch.int32).pin_memory()    # offsets (i64) | hw (i32)
        self.meta_d = torch.empty(max_batch * 4, dtype=torch.int32, device=self.device)
        self.copied = None                      # event: staging buffers free for reuse

    def __call__(self, images, out=None):
        n = len(images)
        if n == 0 or n > self.max_batch:
            raise ValueError('Preprocessor: batch of %d images (max %d)' % (n, self.max_batch))
        if self.copied is not None:
            self.copied.synchronize()
        offs = self.meta_h[:2 * self.max_batch].view(torch.int64)
        hw = self.meta_h[2 * self.max_batch:]
        pos = 0
        for i, img in enumerate(images):
            a = np.ascontiguousarray(img)
            if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
                raise ValueError('Preprocessor: image %d is %s %s, expected uint8 HxWx3' % (i, a.dtype, a.shape))
            if pos + a.size > self.cap:
                raise _lib.CtdetError('Preprocessor: staging capacity %d bytes exceeded' % self.cap)
            self.stage[pos:pos + a.size] = torch.from_numpy(a.reshape(-1))
            offs[i], hw[2 * i], hw[2 * i + 1] = pos, a.shape[0], a.shape[1]
            pos += (a.size + 15) // 16 * 16
        self.dev[:pos].copy_(self.stage[:pos], non_blocking=True)
        self.meta_d.copy_(self.meta_h, non_blocking=True)
        self.copied = torch.cuda.Event()
        self.copied.record()
        if out is None:
            out = torch.empty(n, 3, self.size, self.size, device=self.device, dtype=torch.float32)
        offs_d = self.meta_d[:2 * self.max_batch]
        hw_d = self.meta_d[2 * self.max_batch:]
        check(lib().ct_preproc_resize(_dev(self.dev, 'src', torch.uint8), _dev(offs_d, 'offsets', torch.int32),
                                      _dev(hw_d, 'hw', torch.int32), n,
                                      self.size, C.cast(self.means, C.c_void_p), _dev(out, 'out'), _stream()),
              'ct_preproc_resize')
        return out


class AugPlan(C.Structure):
    """One image's augmentation decisions (csrc/ct_preproc.hip AugPlan, 96 bytes)."""
    _fields_ = [('src_off', C.c_longlong), ('H', C.c_int), ('W', C.c_int),
                ('crop_l', C.c_int), ('crop_t', C.c_int), ('crop_w', C.c_int), ('crop_h', C.c_int),
                ('exp_w', C.c_int), ('exp_h', C.c_int), ('exp_left', C.c_int), ('exp_top', C.c_int),
                ('mirror', C.c_int), ('interp', C.c_int), ('flags', C.c_int), ('hue_delta', C.c_int),
                ('beta', C.c_float), ('alpha', C.c_float), ('sat_alpha', C.c_float), ('fill', C.c_float * 3),
                ('pad0', C.c_float), ('pad1', C.c_float)]


assert C.sizeof(AugPlan) == 96


class Augmenter:
    """Batched training-time augmentation (data/data_augment.py:164-221) on the device: the host hands over the
    uint8 images and one AugPlan per image (the random decisions); ONE launch writes float32 [B,3,S,S]."""

    def __init__(self, size, means, device, max_batch=32, max_pixels=512 * 512):
        self.size, self.device, self.max_batch = size, torch.device(device), max_batch
        self.means_f = [float(m) for m in means]
        self.means = (C.c_float * 3)(*self.means_f)
        self.cap = max_batch * max_pixels * 3
        self.stage = torch.empty(self.cap, dtype=torch.uint8).pin_memory()
        self.dev = torch.empty(self.cap, dtype=torch.uint8, device=self.device)
        self.plan_h = torch.empty(max_batch * 96, dtype=torch.uint8).pin_memory()
        self.plan_d = torch.empty(max_batch * 96, dtype=torch.uint8, device=self.device)
        self.copied = None

    def __call__(self, images, plans, out=None):
        n = len(images)
        if n == 0 or n > self.max_batch or len(plans) != n:
            raise ValueError('Augmenter: batch of %d images / %d plans (max %d)' % (n, len(plans), self.max_batch))
        if self.copied is not None:
            self.copied.synchronize()
        need = sum(int(np.asarray(img).size) for img in images)
        if need > self.cap:                 # any image size, like the reference (COCO is 640x480): grow, never refuse
            self.cap = int(need * 1.25)
            self.stage = torch.empty(self.cap, dtype=torch.uint8).pin_memory()
            self.dev = torch.empty(self.cap, dtype=torch.uint8, device=self.device)
        recs = (AugPlan * n)()
        pos = 0
        for i, (img, p) in enumerate(zip(images, plans)):
            a = np.ascontiguousarray(img)
            if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3 or (a.shape[0], a.shape[1]) != (p['H'], p['W']):
                raise ValueError('Augmenter: image %d is %s %s, plan says %dx%d' % (i, a.dtype, a.shape, p['H'], p['W']))
            if pos + a.size > self.cap:
                raise _lib.CtdetError('Augmenter: staging capacity %d bytes exceeded' % self.cap)
            self.stage[pos:pos + a.size] = torch.from_numpy(a.reshape(-1))
            r = recs[i]
            r.src_off, r.H, r.W = pos, p['H'], p['W']
            r.crop_l, r.crop_t, r.crop_w, r.crop_h = p['crop']
            r.exp_w, r.exp_h, r.exp_left, r.exp_top = p['exp']
            r.mirror, r.interp, r.flags, r.hue_delta = p['mirror'], p['interp'], p['flags'], p['hue']
            r.beta, r.alpha, r.sat_alpha = p['beta'], p['alpha'], p['sat']
            for c in range(3):
                r.fill[c] = self.means_f[c]
            pos += (a.size + 15) // 16 * 16
        self.plan_h[:n * 96] = torch.frombuffer(bytearray(bytes(recs)), dtype=torch.uint8)
        self.dev[:pos].copy_(self.stage[:pos], non_blocking=True)
        self.plan_d[:n * 96].copy_(self.plan_h[:n * 96], non_blocking=True)
        self.copied = torch.cuda.Event()
        self.copied.record()
        if out is None:
            out = torch.empty(n, 3, self.size, self.size, device=self.device, dtype=torch.float32)
        check(lib().ct_preproc_augment(_dev(self.dev, 'src', torch.uint8), _dev(self.plan_d, 'plans', torch.uint8), n,
                                       self.size, C.cast(self.means, C.c_void_p), _dev(out, 'out'), _stream()),
              'ct_preproc_augment')
        return out


def mixup_blend(img1, img2, lambd):
    """img1 * lambd + img2 * (1 - lambd) per image (data/voc0712.py:262); lambd: float or [B] tensor."""
    B = img1.shape[0]
    lam = torch.as_tensor(lambd, dtype=torch.float32, device=img1.device).expand(B).contiguous()
    out = torch.empty_like(img1)
    check(lib().ct_mixup_blend(_dev(img1, 'img1'), _dev(img2, 'img2'), _dev(lam, 'lambd'), B, img1.numel() // B,
                               _dev(out, 'out'), _stream()), 'ct_mixup_blend')
    return out


# ------------------------------------------------------------------ pooling / attention
def maxpool2d(x, k, stride, pad=0, ceil_mode=False):
    B, Cn, H, W = x.shape

    def osz(n):
        num = n + 2 * pad - k
        o = (-(-num // stride) if ceil_mode else num // stride) + 1
        if ceil_mode and (o - 1) * stride >= n + pad:
            o -= 1
        return o
    OH, OW = osz(H), osz(W)
    out = torch.empty(B, Cn, OH, OW, device=x.device, dtype=torch.float32)
    check(lib().ct_maxpool2d_fwd(_dev(x, 'x'), _dev(out, 'out'), B * Cn, H, W, OH, OW, k, stride, pad,
                                 _stream()), 'ct_maxpool2d_fwd')
    return out


def _ctx_prm(params, d, T, incre):
    prm = _lib.CtxParams()
    for k in ('theta_w', 'theta_b', 'phi_w', 'phi_b', 'g_w', 'g_b', 'wz', 'obj_w'):
        setattr(prm, k, _dev(params[k], k).value)
    if incre:
        prm.fc_w = _dev(params['fc_w'], 'fc_w').value
        prm.fc_b = _dev(params['fc_b'], 'fc_b').value
    prm.scale = float(params['scale'])
    prm.d, prm.t = d, T
    return prm


def ctx_attention(conf, pool, params, setting_incre=False, out=None, ws=None):
    """models/RFB_Net_vgg.py:253-271.  params: dict of device tensors theta_w, theta_b, phi_w, phi_b,
    g_w, g_b, wz, obj_w[, fc_w, fc_b] and float `scale`.  `out` / `ws`: caller-owned output and workspace
    (ctx_attention_buffers) so a captured / allocation-free forward can reuse them."""
    B, P, d = conf.shape
    M = pool.shape[1]
    T = params['obj_w'].shape[0]
    prm = _ctx_prm(params, d, T, setting_incre)
    if out is None:
        out = torch.empty(B, P, (d if setting_incre else 0) + T, device=conf.device, dtype=torch.float32)
    ws_bytes = lib().ct_ctx_attention_workspace_bytes(B, P, M, d)
    if ws is None:
        ws = torch.empty(ws_bytes, device=conf.device, dtype=torch.uint8)
    check(lib().ct_ctx_attention_fwd(_dev(conf, 'conf'), _dev(pool, 'pool'), B, P, M, C.byref(prm),
                                     _dev(out, 'out'), _dev(ws, 'ws', torch.uint8), ws_bytes, _stream()),
          'ct_ctx_attention_fwd')
    return out


def ctx_attention_buffers(B, P, M, d, T, setting_incre, device):
    """(out, ws) for ctx_attention with these sizes."""
    out = torch.empty(B, P, (d if setting_incre else 0) + T, device=device, dtype=torch.float32)
    ws = torch.empty(lib().ct_ctx_attention_workspace_bytes(B, P, M, d), device=device, dtype=torch.uint8)
    return out, ws


class CtxTrainer:
    """Forward-with-saved-rows + backward of the Context-Transformer block for one (B, P, M, d, T)
    (train.py:222-229 differentiates models/RFB_Net_vgg.py:253-271 through autograd).  Owns the
    workspaces; gradients come back in fresh tensors keyed like `params`."""
    KEYS = ('theta_w', 'theta_b', 'phi_w', 'phi_b', 'g_w', 'g_b', 'wz', 'obj_w')

    def __init__(self, B, P, M, d, T, incre, device):
        self.B, self.P, self.M, self.d, self.T, self.incre = B, P, M, d, T, bool(incre)
        self.device = torch.device(device)
        L = lib()
        self.ws_bytes = max(L.ct_ctx_attention_workspace_bytes(B, P, M, d),
                            L.ct_ctx_attention_bwd_workspace_bytes(B, P, M))
        self.ws = torch.empty(self.ws_bytes, device=self.device, dtype=torch.uint8)
        self.saved_bytes = L.ct_ctx_attention_saved_bytes(B, P)
        self.saved = torch.empty(self.saved_bytes, device=self.device, dtype=torch.uint8)
        self.keys = self.KEYS + (('fc_w', 'fc_b') if self.incre else ())
        self.dconf = torch.empty(B, P, d, device=self.device)
        self.dpool = torch.empty(B, M, d, device=self.device)

    def forward(self, conf, pool, params):
        prm = _ctx_prm(params, self.d, self.T, self.incre)
        out = torch.empty(self.B, self.P, (self.d if self.incre else 0) + self.T, device=self.device)
        check(lib().ct_ctx_attention_fwd_train(_dev(conf, 'conf'), _dev(pool, 'pool'), self.B, self.P, self.M,
                                               C.byref(prm), _dev(out, 'out'), _dev(self.saved, 'saved', torch.uint8),
                                               self.saved_bytes, _dev(self.ws, 'ws', torch.uint8), self.ws_bytes,
                                               _stream()), 'ct_ctx_attention_fwd_train')
        return out

    def backward(self, conf, pool, params, dout):
        """-> (dconf [B,P,d] without the pooled path, dpool [B,M,d], {key: grad})."""
        prm = _ctx_prm(params, self.d, self.T, self.incre)
        grads = {k: torch.empty_like(params[k]) for k in self.keys}
        g = _lib.CtxGrads()
        for k in self.keys:
            setattr(g, k, _dev(grads[k], 'd' + k).value)
        check(lib().ct_ctx_attention_bwd(_dev(conf, 'conf'), _dev(pool, 'pool'), self.B, self.P, self.M, C.byref(prm),
                                         _dev(self.saved, 'saved', torch.uint8), _dev(dout, 'dout'),
                                         _dev(self.dconf, 'dconf'), _dev(self.dpool, 'dpool'), C.byref(g),
                                         _dev(self.ws, 'ws', torch.uint8), self.ws_bytes, _stream()),
              'ct_ctx_attention_bwd')
        return self.dconf, self.dpool, grads


def ctx_pool_bwd(conf_flat, src_base, dpool_flat, dst_base, dconf_flat, B, h, w, ch, k):
    """Adds the pooled-path gradient of one source into the flat conf gradient (element offsets)."""
    check(lib().ct_ctx_pool_bwd(C.c_void_p(_dev(conf_flat, 'conf').value + 4 * src_base), conf_flat.shape[1],
                                C.c_void_p(_dev(dpool_flat, 'dpool').value + 4 * dst_base), dpool_flat.shape[1],
                                C.c_void_p(_dev(dconf_flat, 'dconf').value + 4 * src_base), dconf_flat.shape[1],
                                B, h, w, ch, k, _stream()), 'ct_ctx_pool_bwd')


def device_info(device=0):
    cu, lds = C.c_int(0), C.c_int(0)
    arch = C.create_string_buffer(64)
    check(lib().ct_device_info(device, C.byref(cu), C.byref(lds), arch, 64), 'ct_device_info')
    return dict(cu_count=cu.value, lds_bytes_per_cu=lds.value, arch=arch.value.decode())
