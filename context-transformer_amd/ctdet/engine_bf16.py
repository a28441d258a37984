"""bf16 channels-last execution of the RFBNet plan (BASELINE.json configs[4]: "bf16 MFMA convs + fp32 NMS").

Same Plan / Runtime / two-stream schedule as the fp32 path (ctdet/engine.py); only the storage and the conv kernel
differ: every activation map is [batch, H, W, C] bfloat16 (C padded to a multiple of 8, so the 8 input channels an
MFMA lane needs at a filter tap are one 16-byte load and `torch.cat` is still a channel offset), convolutions run
`ct_conv2d_bf16_fwd` (v_mfma_f32_32x32x16_bf16, fp32 accumulate + fp32 epilogue), the multibox heads write their
fp32 channels-last outputs exactly as the fp32 kernels do, and everything after them (softmax, decode, NMS,
Context-Transformer block) is the unchanged fp32 code.  What models/RFB_Net_vgg.py:219-248 computes, at bf16
activation precision: select with `net.conv_dtype = 'bf16'` (or CTDET_DTYPE=bf16) before the first forward.
"""
import ctypes as C
import os

import torch

from . import _lib
from .engine import HipBackend


class HipBackendBF16(HipBackend):
    tune_conv = None                 # one tile shape: nothing to tune, no Winograd routing

    def alloc(self, shape, dtype=torch.float32):
        if len(shape) == 4 and dtype == torch.float32:          # (batch, C, H, W) of the plan -> NHWC bf16
            b, c, h, w = shape
            return torch.zeros((b, h, w, (c + 7) // 8 * 8), device=self.device, dtype=torch.bfloat16)
        return super().alloc(shape, dtype)

    def load_input(self, xbuf, x):
        b, c, h, w = x.shape
        _lib.check(self.lib.ct_nchw_f32_to_nhwc_bf16(x.data_ptr(), b, c, h * w, xbuf.shape[3], xbuf.data_ptr(),
                                                     self._stream()), 'input -> NHWC bf16')

    def prepare_conv(self, st, bufs, batch):
        lib, rt = self.lib, st.rt
        src = bufs[st.src]
        cin_buf = src.shape[3] if st.src == 'x' else st.cin      # the image is stored with 8 channels (5 zero)
        if st.src_coff % 8 or cin_buf % 8:
            raise _lib.CtdetError('%s: bf16 path needs channel slices in multiples of 8 (offset %d, %d channels)'
                                  % (st.name, st.src_coff, cin_buf))
        mpad = lib.ct_conv_mpad(st.cout)
        rt['wpk16'] = torch.empty(lib.ct_conv_bf16_packed_elems(st.cin, st.cout, st.kh, st.kw), dtype=torch.int16,
                                  device=self.device)
        rt['scale'] = torch.ones(mpad, device=self.device)
        rt['shift'] = torch.zeros(mpad, device=self.device)
        relus = [p.relu for p in st.parts]
        rt['lo'] = None
        if any(relus) and not all(relus):
            lo = torch.zeros(mpad, device=self.device)
            off = 0
            for p in st.parts:
                lo[off:off + p.cout] = 0.0 if p.relu else -float('inf')
                off += p.cout
            rt['lo'] = lo
        rt['mpad'] = mpad
        self.pack_conv(st)
        d = _lib.ConvDesc()
        d.in_ = src.data_ptr()
        d.batch, d.cin, d.h, d.w = batch, cin_buf, st.h, st.w
        d.in_ctot, d.in_coff = src.shape[3], st.src_coff
        d.wpacked, d.scale, d.shift = rt['wpk16'].data_ptr(), rt['scale'].data_ptr(), rt['shift'].data_ptr()
        d.cout = st.cout
        d.kh, d.kw, d.stride, d.pad_h, d.pad_w, d.dil = st.kh, st.kw, st.stride, st.ph, st.pw, st.dil
        d.oh, d.ow = st.oh, st.ow
        if st.segs:
            d.nseg = len(st.segs)
            for g, sg in enumerate(st.segs):
                t = bufs[sg.dst]
                d.seg[g].ptr = t.data_ptr()
                d.seg[g].co_begin, d.seg[g].co_end = sg.co_begin, sg.co_end
                d.seg[g].pix_stride, d.seg[g].img_stride, d.seg[g].base = sg.pix_stride, t.shape[1], sg.base
        else:
            dst = bufs[st.dst]
            assert dst.shape[1] == st.oh and dst.shape[2] == st.ow, (st.name, dst.shape, st.oh, st.ow)
            d.out, d.out_ctot, d.out_coff = dst.data_ptr(), dst.shape[3], st.dst_coff
        if st.res is not None:
            r = bufs[st.res]
            d.res, d.res_ctot, d.res_coff, d.res_scale = r.data_ptr(), r.shape[3], st.res_coff, st.res_scale
        d.relu = int(all(relus))
        d.lo = rt['lo'].data_ptr() if rt['lo'] is not None else None
        npix = batch * st.oh * st.ow
        if int(os.environ.get('CTDET_KSPLIT', '1')) and st.cout * npix <= (2 << 20):
            rt['ksws'] = torch.empty(16 * st.cout * npix, device=self.device)     # split-K slabs (small maps)
            d.ksplit, d.ksplit_ws, d.ksplit_ws_floats = -1, rt['ksws'].data_ptr(), rt['ksws'].numel()
        rt['desc'] = d
        rt['wino_ok'] = False

    def pack_conv(self, st):
        rt, lib = st.rt, self.lib
        n = len(st.parts)
        ws = [p.weight.detach() for p in st.parts]
        for wt in ws:
            if not (wt.is_cuda and wt.is_contiguous() and wt.dtype == torch.float32):
                raise _lib.CtdetError('%s: parameters must be contiguous fp32 on the HIP device' % st.name)
        ptrs = (C.c_void_p * n)(*[wt.data_ptr() for wt in ws])
        couts = (C.c_int * n)(*[p.cout for p in st.parts])
        _lib.check(lib.ct_conv_pack_weights_bf16(ptrs, couts, n, st.cin, st.kh, st.kw, rt['wpk16'].data_ptr(),
                                                 self._stream()), 'ct_conv_pack_weights_bf16')
        off = 0
        for p in st.parts:
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

    def enable_wino(self, st, on=True):
        if on:
            raise _lib.CtdetError('the bf16 path has no Winograd routing')

    def run_conv(self, st):
        _lib.check(self.lib.ct_conv2d_bf16_fwd(C.byref(st.rt['desc']), self._stream()), st.name)

    def run_pool(self, st, bufs, batch):
        src, dst = bufs[st.src], bufs[st.dst]
        assert src.shape[3] == dst.shape[3]
        _lib.check(self.lib.ct_maxpool2d_nhwc_bf16(src.data_ptr(), dst.data_ptr(), batch, src.shape[3], st.h, st.w,
                                                   st.oh, st.ow, st.k, st.stride, st.pad, self._stream()), st.name)
