"""Training engine: forward with batch-statistics BatchNorm + the full backward pass of the RFBNet
backbone and heads as HIP launches, exposed to autograd as ONE function.

What the reference gets from autograd + cuDNN in `train.py:222-229` (`model(data)` ...
`losses.backward()`) is rebuilt on the same fused launch plan as inference (ctdet.engine.Plan):

  forward   conv (identity epilogue) -> ct_bn_train_stats (batch mean / biased var, running-stat
            update with momentum 0.01) -> ct_bn_train_apply (+ReLU, + `out*scale + shortcut`) for
            BasicConv layers; the fused bias+ReLU conv for the VGG trunk and the heads.
  backward  reverse walk: ct_bn_train_backward / ct_bias_act_backward -> dZ, then per fused conv ONE
            weight-gradient launch (ct_conv2d_wgrad) and ONE data-gradient launch (ct_conv2d_fwd,
            transposed mode, accumulating into the producer's gradient buffer through the residual
            input), ct_maxpool2d_bwd for the pools, ct_head_grad_gather for the head scatter.

  phase 2   the Context-Transformer block on top of the conf head: ct_ctx_pool_fwd +
            ct_ctx_attention_fwd_train forward, ct_ctx_attention_bwd + ct_ctx_pool_bwd backward
            (ops.CtxTrainer), producing the gradient of the raw conf logits the head backward consumes.

Buffers are owned by the runtime and reused every step (call backward before the next forward).
"""
import ctypes as C
import os

import torch

from . import _lib, ops
from .engine import (F4_TILES, H2_TILES, TRACK_TILES, WINO4F_TILES, WINO4S_TILES, ConvPart, ConvStep, HipBackend, Plan, Runtime,
                     apply_tuned, operand_form_h2, run_on_streams)


class _StepState:
    pass


class TrainRuntime:
    def __init__(self, net, batch, backend):
        self.net, self.batch, self.be = net, batch, backend
        self.lib = backend.lib
        self.plan = Plan(net, batch)
        al = backend.alloc
        self.bufs = {n: al((batch,) + tuple(s)) for n, s in self.plan.buf_shapes.items()}
        self.grads = {n: al((batch,) + tuple(s)) for n, s in self.plan.buf_shapes.items() if n != 'x'}
        self.state = {}
        self._batched_packs = os.environ.get('CTDET_PACK_BATCH', '1') != '0'
        self._pack_table, self._pack_ptrs, self._pack_counts = None, None, (0, 0)
        self.params = []            # ordered parameters that receive gradients
        self._pindex = {}
        self.ctx = None
        if self.plan.ctx:
            C_ = net.num_classes
            self.P = self.bufs['conf'].shape[1] // C_
            self.ctx = ops.CtxTrainer(batch, self.P, self.plan.M, C_, net.OBJ_Target.weight.shape[0],
                                      net.setting == 'incre', backend.device)
            self.ctx_params = dict(obj_w=net.OBJ_Target.weight, wz=net.Wz, theta_w=net.theta.weight,
                                   theta_b=net.theta.bias, phi_w=net.phi.weight, phi_b=net.phi.bias,
                                   g_w=net.g.weight, g_b=net.g.bias)
            if net.setting == 'incre':
                self.ctx_params.update(fc_w=net.fc_base.weight, fc_b=net.fc_base.bias)
            for prm in self.ctx_params.values():
                self._reg(prm)
        wino_ws = 0
        w4s_ws = 0              # bytes: workspace of the three-kernel Winograd data gradients
        wg4s_ws = 0             # bytes: workspace of the three-kernel Winograd weight gradients
        # CTDET_TRAIN_WINO4=0 keeps forward and data-gradient convolutions on F(2x2,3x3) where the table says F(4x4,3x3)
        wino4 = os.environ.get('CTDET_TRAIN_WINO4', '1') != '0'
        # The Winograd launches of the step (forward and data gradients) on the f16x2 operand form (csrc/ct_f16x2.h) wherever the
        # table's bf16x3 tile has that twin, under the inference runtime's batch rule (engine.operand_form_h2); CTDET_TRAIN_H2=0
        # keeps bf16x3.  Direct layers stay on bf16x3: their inputs come from the BatchNorm kernels, which do not track maxima.
        self.h2 = os.environ.get('CTDET_TRAIN_H2', '1') != '0' and operand_form_h2(net, batch)[0]
        backend.h2, backend.h2_direct = self.h2, False
        if self.plan.ctx:
            backend.w4f_max_cin = int(os.environ.get('CTDET_TRAIN_CTX_W4F_MAX_CIN', '128'))      # engine.apply_tuned
        for st in self.plan.steps:
            if st.kind != 'conv':
                continue
            s = _StepState()
            self.state[st.name] = s
            s.is_bn = st.parts[0].bn is not None
            assert all((p.bn is not None) == s.is_bn for p in st.parts), st.name
            ctot = st.cout
            # channel count of the dZ buffer.  A multibox head has 4 k + C k + k output channels (156 for six anchors and 20
            # classes): not a multiple of 8 / 16, so its data gradient (a 3x3 convolution WITH dZ's channels as input) fitted
            # neither the Winograd nor the bf16x3 kernel and ran on the fp32 implicit GEMM (2 x 450-770 us per step).  dZ
            # is padded with zero channels instead (ct_head_grad_gather writes 0 outside its segments; the packed
            # data-gradient weights get a zero part of as many rows).
            s.zc = (ctot + 15) // 16 * 16 if (st.segs and st.src != 'x' and (st.kh, st.kw, st.stride, st.dil) == (3, 3, 1, 1)) else ctot
            s.zero_w = torch.zeros(s.zc - ctot, st.cin, st.kh, st.kw, device=backend.device) if s.zc > ctot else None
            s.dz = al((batch, s.zc, st.oh, st.ow))
            s.dw = al((ctot, st.cin, st.kh, st.kw))
            if s.is_bn:
                # conv with identity epilogue into a dense Z buffer
                zname = 'z.' + st.name
                self.bufs[zname] = al((batch, ctot, st.oh, st.ow))
                s.zstep = ConvStep(zname, [ConvPart(p.weight, None, None, False) for p in st.parts], st.cin,
                                   st.kh, st.kw, st.stride, st.ph, st.pw, st.dil, st.src, st.src_coff, st.h,
                                   st.w, zname, 0)
                backend.prepare_conv(s.zstep, self.bufs, batch)
                apply_tuned(backend, s.zstep, batch, wino4=wino4)
                s.fwd = s.zstep
                s.mean = [al((p.cout,)) for p in st.parts]
                s.var = [al((p.cout,)) for p in st.parts]
                s.dgamma = [al((p.cout,)) for p in st.parts]
                s.dbeta = [al((p.cout,)) for p in st.parts]
                s.scratch = [al((2 * p.cout,), torch.float64) for p in st.parts]     # sliced BN reductions
            else:
                backend.prepare_conv(st, self.bufs, batch)
                apply_tuned(backend, st, batch, wino4=wino4)
                s.fwd = st
                s.dbias = [al((p.cout,)) for p in st.parts]
            # data-gradient launch (not needed for the image itself)
            s.dgrad = None
            if st.src != 'x':
                zc = s.zc
                kpad = self.lib.ct_conv_kpad(zc, st.kh, st.kw)
                mpad = self.lib.ct_conv_mpad(st.cin)
                s.wpk_d = al((kpad, mpad))
                s.ones = torch.ones(mpad, device=backend.device)
                s.zeros = torch.zeros(mpad, device=backend.device)
                d = _lib.ConvDesc()
                d.in_ = s.dz.data_ptr()
                d.batch, d.cin, d.h, d.w, d.in_ctot, d.in_coff = batch, zc, st.oh, st.ow, zc, 0
                d.wpacked, d.scale, d.shift = s.wpk_d.data_ptr(), s.ones.data_ptr(), s.zeros.data_ptr()
                d.cout, d.m_pad, d.k_pad = st.cin, mpad, kpad
                d.kh, d.kw, d.stride, d.pad_h, d.pad_w, d.dil = st.kh, st.kw, st.stride, st.ph, st.pw, st.dil
                d.oh, d.ow = st.h, st.w
                g = self.grads[st.src]
                d.out, d.out_ctot, d.out_coff = g.data_ptr(), g.shape[1], st.src_coff
                d.res_ctot, d.res_coff, d.res_scale = g.shape[1], st.src_coff, 1.0
                d.transposed = 1
                if int(os.environ.get('CTDET_KSPLIT', '1')) and st.cin * batch * st.h * st.w <= (2 << 20):
                    s.ksws_d = al((16 * st.cin * batch * st.h * st.w,))      # split-K slabs (small maps)
                    d.ksplit, d.ksplit_ws, d.ksplit_ws_floats = -1, s.ksws_d.data_ptr(), s.ksws_d.numel()
                s.dgrad = d
                s.kpad_d, s.mpad_d = kpad, mpad
                # 3x3 / stride 1 / pad 1 layers: the data gradient is itself such a convolution (channels
                # swapped, taps rotated) -> Winograd kernel on dY with ct_conv_pack_weights_wino_dgrad
                s.dgrad_wino = None
                if (st.kh, st.kw, st.stride, st.dil, st.ph, st.pw) == (3, 3, 1, 1, 1, 1) and zc % 8 == 0 \
                        and st.oh * st.ow >= 19 * 19:
                    w2 = _lib.ConvDesc()
                    C.memmove(C.byref(w2), C.byref(d), C.sizeof(d))
                    w2.transposed = 0
                    w2.kh = w2.kw = 3
                    if self.lib.ct_conv_wino_supported(C.byref(w2)):
                        s.dgrad_wino = w2
                        # F(4x4,3x3) where the forward launch of this layer uses it (same map, channels swapped) and on
                        # the multibox heads from 19x19 maps up (their forward launch is a bf16x3 / F(2x2) one chosen for
                        # cout = 156; the data gradient has cout = the source's channel count)
                        w4_ok = bool(self.lib.ct_conv_wino4_supported(C.byref(w2)))
                        s.dgrad_tile = 4 if w4_ok and (s.fwd.rt.get('wino') in F4_TILES or
                                                       (wino4 and st.segs and st.oh * st.ow >= 361)) else 2
                        # ... and its three-kernel bf16x3 form (tile 44) where the forward launch runs that one
                        # (CTDET_TRAIN_W4S=0 keeps the fused kernel); V / M workspace shared by all data gradients
                        if s.fwd.rt.get('wino') in WINO4S_TILES and os.environ.get('CTDET_TRAIN_W4S', '1') != '0' and \
                                self.lib.ct_conv_wino4s_supported(C.byref(w2)):
                            s.dgrad_tile = 47 if self.h2 else 44
                            size = self.lib.ct_conv_wino4s_h2_packed_bytes if self.h2 else self.lib.ct_conv_wino4s_packed_bytes
                            s.U_d = al(((size(zc, st.cin) + 3) // 4,))
                            w4s_ws = max(w4s_ws, self.lib.ct_conv_wino4s_workspace_bytes(C.byref(w2)))
                        elif s.fwd.rt.get('wino') in WINO4F_TILES and os.environ.get('CTDET_TRAIN_W4F', '1') != '0' and \
                                zc <= (getattr(backend, 'w4f_max_cin', None) or 1 << 30) and \
                                self.lib.ct_conv_wino4f_supported(C.byref(w2)):
                            # ... and the fused bf16x3 F(4x4,3x3) kernel (tile 46) where the forward launch runs it: the narrow
                            # layers on the big maps, whose data gradients were 7 launches x 1.13 ms of the 37.8 ms step on the
                            # fp32 kernel (profiles/r05_train_kernel_stats.md)
                            # (f16x2, tile 48: where dZ comes from ct_bias_act_backward_amax, which leaves the maxima the fused
                            # kernel needs -- the VGG trunk; a BatchNorm layer's dZ has none: bf16x3)
                            s.dgrad_tile = 48 if self.h2 and not s.is_bn and not st.segs else 46
                            size = self.lib.ct_conv_wino4f_h2_packed_bytes if s.dgrad_tile == 48 else self.lib.ct_conv_wino4f_packed_bytes
                            s.U_d = al(((size(zc, st.cin) + 3) // 4,))
                        else:
                            sizeof = self.lib.ct_conv_wino4_packed_floats if s.dgrad_tile == 4 else self.lib.ct_conv_wino_packed_floats
                            s.U_d = al((sizeof(zc, st.cin),))
                # dilated 3x3 layers (pad = dilation) whose forward launch runs the three-kernel form: their data gradient is
                # the same dilated convolution with channels swapped and taps rotated -> the same kernels (tiles on the
                # dilation sub-lattices); the dilated output transform has no accumulate, so a source gradient that was already
                # written goes through a scratch tensor (does not happen in these networks)
                elif (st.kh, st.kw, st.stride) == (3, 3, 1) and st.dil > 1 and st.ph == st.pw == st.dil and zc % 16 == 0 and \
                        s.fwd.rt.get('wino') in WINO4S_TILES and os.environ.get('CTDET_TRAIN_W4S', '1') != '0' and \
                        os.environ.get('CTDET_TRAIN_W4S_DIL', '1') != '0':
                    w2 = _lib.ConvDesc()
                    C.memmove(C.byref(w2), C.byref(d), C.sizeof(d))
                    w2.transposed = 0
                    w2.ksplit, w2.ksplit_ws, w2.ksplit_ws_floats = 0, None, 0
                    if self.lib.ct_conv_wino4s_supported(C.byref(w2)):
                        s.dgrad_wino = w2
                        s.dgrad_tile = 47 if self.h2 else 44
                        size = self.lib.ct_conv_wino4s_h2_packed_bytes if self.h2 else self.lib.ct_conv_wino4s_packed_bytes
                        s.U_d = al(((size(zc, st.cin) + 3) // 4,))
                        w4s_ws = max(w4s_ws, self.lib.ct_conv_wino4s_workspace_bytes(C.byref(w2)))
            # direct data gradients on the bf16 matrix pipe (bf16x3, ct_conv2d_x3_fwd transposed): every layer without
            # a Winograd data gradient whose channel counts fit the k-step; CTDET_X3=0 keeps ct_conv2d_fwd
            s.dgrad_x3 = None
            if s.dgrad is not None and getattr(s, 'dgrad_wino', None) is None and st.stride <= 2 and \
                    os.environ.get('CTDET_X3', '1') != '0' and s.zc % 16 == 0 and s.zc >= 32:
                npix = batch * st.h * st.w
                tiles128 = -(-st.cin // 128) * -(-npix // 128)
                cfg = 0 if tiles128 >= 512 else (3 if s.zc % 32 == 0 and -(-st.cin // 64) * -(-npix // 128) < 384 else 1)
                bk = self.lib.ct_conv_x3_config_bk(cfg)
                s.dgrad_x3 = cfg
                s.wx3_d = al((self.lib.ct_conv_x3_packed_bytes(s.zc, st.cin, st.kh, st.kw, bk),), torch.uint8)
            # weight-gradient descriptor = forward geometry on the forward input
            w = _lib.ConvDesc()
            src = self.bufs[st.src]
            w.in_ = src.data_ptr()
            w.batch, w.cin, w.h, w.w, w.in_ctot, w.in_coff = batch, st.cin, st.h, st.w, src.shape[1], st.src_coff
            w.cout = ctot
            w.kh, w.kw, w.stride, w.pad_h, w.pad_w, w.dil = st.kh, st.kw, st.stride, st.ph, st.pw, st.dil
            w.oh, w.ow = st.oh, st.ow
            s.wgrad = w
            # 3x3 / stride 1 / pad 1: Winograd F(3x3, 2x2) weight gradient; one workspace shared by all layers
            # (stream ordered).  Maps below 10x10 stay on the direct kernel (tile padding costs more than the
            # transform saves there); CTDET_WGRAD_WINO=0 keeps the direct kernel everywhere.
            s.wgrad_wino = bool(int(os.environ.get('CTDET_WGRAD_WINO', '1'))) and st.oh * st.ow >= 100 and \
                bool(self.lib.ct_conv_wgrad_wino_supported(C.byref(w)))
            # F(3x3, 4x4) from 19x19 maps up (15-20 % faster than F(3x3, 2x2) there, tools/wgrad_probe.py; slower on
            # 10x10); CTDET_WGRAD_WINO4=0 keeps F(3x3, 2x2)
            s.wgrad_tile = 4 if s.wgrad_wino and st.oh * st.ow >= 361 and \
                os.environ.get('CTDET_WGRAD_WINO4', '1') != '0' else 2
            # the three-kernel bf16x3 form (ct_conv2d_wgrad_wino4s) for the wide layers: from CTDET_WGRAD_W4S_MIN_CIN input
            # channels up (default 256; 0 = never) where cin x cout >= 2^17 -- conv4_x, conv5_x, the 19x19 RFB layers
            # (profiles/r04_wgrad_probe.txt: 512 -> 512 @38x38 778 -> 503 us, 512 -> 512 @19x19 253 -> 184, 256 -> 512 @38x38
            # 437 -> 366; the multibox heads (cout 126..156: 238 -> 353) and 256 -> 256 @75x75 (750 -> 804) stay fused).  Its
            # workspace (E, V, dU slabs) is shared: weight gradients run in stream order.
            w4s_min = int(os.environ.get('CTDET_WGRAD_W4S_MIN_CIN', '256') or 0)
            # dilated 3x3 layers (conv6: 512 -> 1024, dilation 6) have no fused Winograd weight gradient; the three-kernel form
            # takes them with the tiles on the dilation sub-lattices, under the same size rule
            dilated = (st.kh, st.kw, st.stride) == (3, 3, 1) and st.dil > 1 and st.ph == st.pw == st.dil and \
                os.environ.get('CTDET_TRAIN_W4S_DIL', '1') != '0'
            # dilated layers: the alternative is the direct fp32 kernel, so the rule is looser -- conv6 and the 256-channel RFB
            # branches (same-box A/B of the step: 2^17 / 2^16 / 2^14 with 128 channels: 38.2-40.1 / 37.7-38.2 / 38.4-38.5 ms)
            dil_prod = int(os.environ.get('CTDET_WGRAD_W4S_DIL_PROD', str(1 << 16)))
            dil_cin = int(os.environ.get('CTDET_WGRAD_W4S_DIL_CIN', '256'))
            if ((s.wgrad_wino and s.wgrad_tile == 4 and st.cin >= w4s_min and st.cin * ctot >= (1 << 17)) or
                    (dilated and st.cin >= dil_cin and st.cin * ctot >= dil_prod)) and w4s_min and \
                    self.lib.ct_conv_wgrad_wino4s_supported(C.byref(w)):
                s.wgrad_wino = True
                s.wgrad_tile = 44
                wg4s_ws = max(wg4s_ws, int(self.lib.ct_conv_wgrad_wino4s_workspace_bytes(C.byref(w))))
            if s.wgrad_wino:
                size = self.lib.ct_conv_wgrad_wino4_workspace_bytes if s.wgrad_tile in (4, 44) else \
                    self.lib.ct_conv_wgrad_wino_workspace_bytes
                s.wgrad_ws_bytes = int(size(C.byref(w))) if s.wgrad_tile != 44 else 64
                wino_ws = max(wino_ws, s.wgrad_ws_bytes)
            for p in st.parts:
                self._reg(p.weight)
                if p.bn is not None:
                    self._reg(p.bn.weight)
                    self._reg(p.bn.bias)
                elif p.bias is not None:
                    self._reg(p.bias)
        # Accumulation buffers the library would otherwise zero with one small memset per launch (~150 per step): the
        # BatchNorm reduction scratch (both passes), the Winograd weight-gradient workspaces (one per layer instead of a
        # shared one) and -- through the gradient arena below -- the bias / split weight gradients.  CTDET_PREZERO=0 keeps
        # the per-launch memsets.
        self.prezero = os.environ.get('CTDET_PREZERO', '1') != '0'
        self.wgrad_ws = al((max(wino_ws // 4, 1),))
        self.dgrad_ws4s = torch.empty(max(w4s_ws, 1), device=backend.device, dtype=torch.uint8)
        self.wgrad_ws4s = torch.empty(max(wg4s_ws, 1), device=backend.device, dtype=torch.uint8)
        if self.prezero:
            bn_floats = sum(t.numel() for s_ in self.state.values() for t in getattr(s_, 'scratch', []))
            self.bn_scratch = al((max(bn_floats, 1),), torch.float64)
            o = 0
            for s_ in self.state.values():
                for i, t in enumerate(getattr(s_, 'scratch', [])):
                    s_.scratch[i] = self.bn_scratch[o:o + t.numel()]
                    o += t.numel()
            sizes = [(s_, s_.wgrad_ws_bytes // 4) for s_ in self.state.values() if s_.wgrad_wino]
            self.wgrad_ws_all = al((max(sum((n + 63) // 64 * 64 for _, n in sizes), 1),))
            o = 0
            for s_, n in sizes:
                s_.wgrad_ws = self.wgrad_ws_all[o:o + n]
                o += (n + 63) // 64 * 64
        self.backend = backend
        Runtime._build_schedule(self)           # forward: Norm branch / heads on a side stream (CTDET_STREAMS)
        # the forward launches of the three-kernel Winograd form share ONE V / M workspace per stream of that schedule
        # (HipBackend.ws_pool); the launch object of a BatchNorm layer is its zstep, not the plan step
        for i, st in enumerate(self.plan.steps):
            if st.kind == 'conv':
                self.state[st.name].fwd.rt['ws_key'] = self.sid[i] if self.side is not None else 0
        backend.ws_rebuild([self.state[st.name].fwd for st in self.plan.steps if st.kind == 'conv'])
        self._wire_absmax()
        # MaxPool2d(2, 2) whose input only the pool reads, under a convolution with bias + ReLU and no BatchNorm (conv1_2, conv2_2,
        # conv3_3): the pool's backward and the convolution's bias / ReLU backward are ONE pass (ct_maxpool2x2_bias_relu_bwd) -- the
        # gradient of the pool's input is never materialised.  CTDET_TRAIN_FUSE_POOL_BWD=0 keeps the two kernels.
        self._pool_bwd = {}
        if os.environ.get('CTDET_TRAIN_FUSE_POOL_BWD', '1') != '0':
            steps = self.plan.steps
            for ps in steps:
                if ps.kind != 'pool' or (ps.k, ps.stride, ps.pad) != (2, 2, 0):
                    continue
                prods = [st for st in steps if st.kind == 'conv' and st.dst == ps.src and not st.segs]
                others = [st for st in steps if st is not ps and (getattr(st, 'src', None) == ps.src or getattr(st, 'res', None) == ps.src)]
                if len(prods) != 1 or others:
                    continue
                pr = prods[0]
                s_ = self.state[pr.name]
                if not s_.is_bn and len(pr.parts) == 1 and pr.parts[0].relu and pr.parts[0].bias is not None and pr.dst_coff == 0 and \
                        pr.cout == ps.ch and s_.zc == pr.cout and pr.res is None and self.grads[ps.dst].shape[1] == ps.ch:
                    self._pool_bwd[ps.name] = pr
        # weight gradients beside the data-gradient chain (CTDET_TRAIN_STREAMS=1 keeps everything on the caller's stream).  The
        # stream is the forward pass's side stream: the two are never busy at the same time, and a training step with ONE side
        # stream can be captured as a hipGraph (a second one crashes hipStreamEndCapture on ROCm 7.2)
        self.wg_stream = (self.side if self.side is not None else torch.cuda.Stream(backend.device)) \
            if int(os.environ.get('CTDET_TRAIN_STREAMS', '2')) > 1 else None
        for s_ in self.state.values():
            s_.ev_dz = torch.cuda.Event()
        self._bns = [p.bn for st in self.plan.steps if st.kind == 'conv' for p in st.parts if p.bn is not None]
        self._nbt = [bn.num_batches_tracked for bn in self._bns]
        # Gradient arena: ONE flat fp32 buffer in production order.  The weight / bias / BatchNorm gradient kernels
        # write straight into their slice (no per-parameter copies), the bucketed all-reduce runs on slices of it,
        # and backward() hands autograd views of a single snapshot.
        order = self.production_order()
        assert len({id(q) for q in order}) == len(order) == len(self.params), 'parameter used by two plan steps'
        self._prod_index = {id(q): i for i, q in enumerate(order)}
        self._arena_off = [0]
        for q in order:
            self._arena_off.append(self._arena_off[-1] + q.numel())
        self.arena = al((max(self._arena_off[-1], 1),))
        for st in self.plan.steps:
            if st.kind != 'conv':
                continue
            s = self.state[st.name]
            for i, p in enumerate(st.parts):
                if s.is_bn:
                    s.dgamma[i], s.dbeta[i] = self._arena_view(p.bn.weight), self._arena_view(p.bn.bias)
                elif p.bias is not None:
                    s.dbias[i] = self._arena_view(p.bias)
            i0 = self._prod_index[id(st.parts[0].weight)]
            assert all(self._prod_index[id(p.weight)] == i0 + k for k, p in enumerate(st.parts))
            a0 = self._arena_off[i0]
            s.dw = self.arena[a0:a0 + s.dw.numel()].view(s.dw.shape)

    def _wire_absmax(self):
        """Maxima of |activation| / |dZ| for the f16x2 launches of the step (engine.Runtime._wire_absmax is the inference version).
        Forward: a buffer has a slot when every step that writes it is a convolution WITHOUT BatchNorm on a tracking kernel (the
        BatchNorm layers' outputs are written by ct_bn_train_apply, which does not track); a pooled buffer shares its source's.
        A fused f16x2 launch (tile 48) without a slot becomes its bf16x3 twin; the three-kernel form (47) takes the maxima
        inside its launch.  Backward: the dZ of a layer without BatchNorm is written by ct_bias_act_backward_amax into a slot of
        its own (s.dz_amax), read by that layer's data-gradient launch."""
        be = self.be
        if not self.h2:
            return
        steps = self.plan.steps
        convs = [st for st in steps if st.kind == 'conv']

        def fwd(st):
            return self.state[st.name].fwd

        def tracks(st):
            if st.kind != 'conv' or st.segs or self.state[st.name].is_bn:
                return False
            w = fwd(st).rt.get('wino')
            return w in TRACK_TILES if w else True
        root = {ps.dst: ps.src for ps in steps if ps.kind == 'pool'}

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
            for st in convs:
                fwd(st).rt['desc'].in_absmax = None
                fwd(st).rt['desc'].out_absmax = None
            slots, fallback = {}, []
            be.slots_used = 0
            for st in convs:
                if fwd(st).rt.get('wino') not in H2_TILES:
                    continue
                b = root_of(st.src)
                ws = writers.get(b, [])
                if ws and all(tracks(w) for w in ws):
                    if b not in slots:
                        slots[b] = be.new_slot(self.batch)
                        for w in ws:
                            fwd(w).rt['desc'].out_absmax = slots[b]
                    fwd(st).rt['desc'].in_absmax = slots[b]
                elif fwd(st).rt.get('wino') == 48:
                    fallback.append(st)
            if not fallback:
                break
            for st in fallback:
                be.enable_wino(fwd(st), tile=46)
        for st in convs:
            s = self.state[st.name]
            s.dz_amax = None
            if getattr(s, 'dgrad_wino', None) is not None and s.dgrad_tile in H2_TILES and not s.is_bn and not st.segs:
                s.dz_amax = be.new_slot(self.batch)
                s.dgrad_wino.in_absmax = s.dz_amax
        self.amax_slots = slots
        self._wired_epoch = be.kernel_epoch

    def _arena_view(self, prm, flat=None):
        i = self._prod_index[id(prm)]
        return (self.arena if flat is None else flat)[self._arena_off[i]:self._arena_off[i + 1]].view(prm.shape)

    def _reg(self, prm):
        if id(prm) not in self._pindex:
            self._pindex[id(prm)] = len(self.params)
            self.params.append(prm)

    def production_order(self):
        """Parameters in the order backward() produces their gradients (heads first)."""
        order = list(self.ctx_params.values()) if self.ctx is not None else []
        for st in reversed(self.plan.steps):
            if st.kind != 'conv':
                continue
            for p in st.parts:
                if p.bn is not None:
                    order += [p.bn.weight, p.bn.bias]
                elif p.bias is not None:
                    order.append(p.bias)
            order += [p.weight for p in st.parts]
        return order

    def enable_grad_sync(self, bucket_bytes=32 << 20, group=None):
        """Data-parallel training: all-reduce (mean) the gradients over the process group in
        large buckets, overlapped with the rest of the backward pass (ctdet.dist.GradBucketer)."""
        from .dist import GradBucketer
        order = self.production_order()
        self.bucketer = GradBucketer([p.numel() for p in order], self.be.device, bucket_bytes, group,
                                     flat=self.arena)
        return self.bucketer

    def _s(self):
        return C.c_void_p(torch.cuda.current_stream(self.be.device).cuda_stream)

    # ------------------------------------------------------------------ weight packing
    def _pack_dgrad(self, st, s):
        """The data-gradient layout of one fused conv from the current weights."""
        wts = [(p.weight.data_ptr(), p.cout) for p in st.parts]
        if s.zero_w is not None:                # the zero channels dZ is padded with (s.zc)
            wts.append((s.zero_w.data_ptr(), s.zero_w.shape[0]))
        n = len(wts)
        ptrs = (C.c_void_p * n)(*[w for w, _ in wts])
        couts = (C.c_int * n)(*[c for _, c in wts])
        if s.dgrad_wino is not None and s.dgrad_tile in H2_TILES:
            if getattr(self, '_recording', False):
                return                  # takes the layer's maximum first: not recordable, batched by the f16x2 list of _repack_all
            pack = self.lib.ct_conv_pack_weights_wino4s_h2_dgrad if s.dgrad_tile == 47 else self.lib.ct_conv_pack_weights_wino4f_h2_dgrad
            _lib.check(pack(ptrs, couts, n, st.cin, s.U_d.data_ptr(), self._s()), st.name + ' pack dgrad (winograd, f16x2)')
        elif s.dgrad_wino is not None:
            pack = {4: self.lib.ct_conv_pack_weights_wino4_dgrad, 44: self.lib.ct_conv_pack_weights_wino4s_dgrad,
                    46: self.lib.ct_conv_pack_weights_wino4f_dgrad}.get(s.dgrad_tile, self.lib.ct_conv_pack_weights_wino_dgrad)
            _lib.check(pack(ptrs, couts, n, st.cin, s.U_d.data_ptr(), self._s()), st.name + ' pack dgrad (winograd)')
        elif s.dgrad_x3 is not None:
            if getattr(self, '_recording', False):
                return                  # not a recordable pack kind: re-issued every step by _repack_all
            _lib.check(self.lib.ct_conv_pack_weights_x3_dgrad(ptrs, couts, n, st.cin, st.kh, st.kw,
                                                              self.lib.ct_conv_x3_config_bk(s.dgrad_x3), s.wx3_d.data_ptr(),
                                                              self._s()), st.name + ' pack dgrad (bf16x3)')
        else:
            _lib.check(self.lib.ct_conv_pack_weights_dgrad(ptrs, couts, n, st.cin, st.kh, st.kw, s.wpk_d.data_ptr(),
                                                           s.mpad_d, s.kpad_d, self._s()), st.name + ' pack dgrad')

    def _repack_all(self):
        """Every optimizer step changes every weight, and every fused conv needs its forward and its data-gradient
        layout: ~130 small pack launches per step.  They are recorded ONCE (ct_pack_record_*) and replayed as two
        batched launches at the start of each forward (the backward pass of the same step reads the same weights).
        CTDET_PACK_BATCH=0 keeps the per-layer launches."""
        if not self._batched_packs:
            return
        # cache key of the recorded lists: the parameter storage AND what each layer's launch reads (kernel choice and
        # packed-weight buffer can change after the first step: enable_x3 / enable_wino / apply_tuned on the shared backend)
        ptrs = [p.data_ptr() for p in self.params]
        for st in self.plan.steps:
            if st.kind == 'conv':
                rt = self.state[st.name].fwd.rt
                x3 = rt.get('x3')
                ptrs.append((st.name, rt.get('wino') or 0, -1 if x3 is None else x3,
                             tuple(sorted((k, v.data_ptr()) for k, v in rt.items()
                                          if k in ('U', 'U4', 'UX', 'U4H', 'U4FH', 'wpk') and v is not None)),
                             tuple(sorted((bk, t.data_ptr()) for bk, t in rt.get('wx3', {}).items()))))
        if self._pack_table is None or ptrs != self._pack_ptrs:
            lib = self.lib
            _lib.check(lib.ct_pack_record_begin(), 'ct_pack_record_begin')
            self._recording = True
            try:
                for st in self.plan.steps:
                    if st.kind != 'conv':
                        continue
                    s = self.state[st.name]
                    # bf16x3 forward layers are split by the batched x3 list below (ct_conv_pack_weights_x3 is not
                    # recordable and would launch right here): only their epilogue is folded
                    # (... and the f16x2 Winograd layouts by the batched list further down: they take the layer's maximum first)
                    self.be.pack_conv(s.fwd, weights=s.fwd.rt.get('x3') is None and s.fwd.rt.get('wino') not in H2_TILES)
                    if s.dgrad is not None:
                        self._pack_dgrad(st, s)
            finally:
                self._recording = False
                nbytes = lib.ct_pack_record_bytes()
                self._pack_table = torch.empty(nbytes, dtype=torch.uint8, device=self.be.device)
                nd, nw = C.c_int(0), C.c_int(0)
                _lib.check(lib.ct_pack_record_end(self._pack_table.data_ptr(), nbytes, C.byref(nd), C.byref(nw), self._s()),
                           'ct_pack_record_end')
            self._pack_counts, self._pack_ptrs = (nd.value, nw.value), ptrs
        _lib.check(self.lib.ct_pack_run(self._pack_table.data_ptr(), self._pack_counts[0], self._pack_counts[1], self._s()),
                   'ct_pack_run')
        # the bf16x3 weight splits (forward + data-gradient launches) are their own list: built once (the arguments never
        # change between steps), replayed as one launch
        if getattr(self, '_x3_list', None) is None or self._x3_list[2] != ptrs:
            items, nbytes = [], self.lib.ct_conv_x3_pack_item_bytes()

            def item(parts, cin, kh, kw, bk, dst, dgrad, zero_w=None):
                wts = [(p.weight.data_ptr(), p.cout) for p in parts]
                if zero_w is not None:
                    wts.append((zero_w.data_ptr(), zero_w.shape[0]))
                n = len(wts)
                wp = (C.c_void_p * n)(*[w for w, _ in wts])
                co = (C.c_int * n)(*[c for _, c in wts])
                buf = (C.c_ubyte * nbytes)()
                _lib.check(self.lib.ct_conv_x3_pack_item(wp, co, n, cin, kh, kw, bk, dst.data_ptr(), dgrad, buf),
                           'ct_conv_x3_pack_item')
                items.append(bytes(buf))
            for st in self.plan.steps:
                if st.kind != 'conv':
                    continue
                s = self.state[st.name]
                if s.fwd.rt.get('x3') is not None:
                    bk = self.be.x3_bk(s.fwd.rt['x3'])
                    item(s.fwd.parts, s.fwd.cin, s.fwd.kh, s.fwd.kw, bk, s.fwd.rt['wx3'][(bk, False)], 0)      # (k-step, f16x2 form): the direct layers of a training step run bf16x3
                if s.dgrad is not None and s.dgrad_x3 is not None:
                    item(st.parts, st.cin, st.kh, st.kw, self.lib.ct_conv_x3_config_bk(s.dgrad_x3), s.wx3_d, 1, s.zero_w)
            table = torch.frombuffer(bytearray(b''.join(items)), dtype=torch.uint8).to(self.be.device) if items else None
            self._x3_list = (table, len(items), ptrs)
        if self._x3_list[1]:
            _lib.check(self.lib.ct_conv_x3_pack_run(self._x3_list[0].data_ptr(), self._x3_list[1], self._s()),
                       'ct_conv_x3_pack_run')
        # the f16x2 Winograd layouts (forward + data-gradient launches): maxima of the weights, then the split -- three launches
        # for the whole list (ct_conv_wino_h2_pack_run)
        if self.h2 and (getattr(self, '_h2_list', None) is None or self._h2_list[2] != ptrs):
            items, nbytes = [], self.lib.ct_conv_wino_h2_pack_item_bytes()

            def h2_item(parts, cin, dgrad, tile, dst, zero_w=None):
                wts = [(p.weight.data_ptr(), p.cout) for p in parts]
                if zero_w is not None:
                    wts.append((zero_w.data_ptr(), zero_w.shape[0]))
                n = len(wts)
                wp = (C.c_void_p * n)(*[w for w, _ in wts])
                co = (C.c_int * n)(*[c for _, c in wts])
                buf = (C.c_ubyte * nbytes)()
                _lib.check(self.lib.ct_conv_wino_h2_pack_item(wp, co, n, cin, dgrad, tile, dst.data_ptr(), buf),
                           'ct_conv_wino_h2_pack_item')
                items.append(bytes(buf))
            for st in self.plan.steps:
                if st.kind != 'conv':
                    continue
                s = self.state[st.name]
                t = s.fwd.rt.get('wino')
                if t in H2_TILES:
                    h2_item(s.fwd.parts, s.fwd.cin, 0, t, s.fwd.rt['U4H' if t == 47 else 'U4FH'])
                if getattr(s, 'dgrad_wino', None) is not None and s.dgrad_tile in H2_TILES:
                    h2_item(st.parts, st.cin, 1, s.dgrad_tile, s.U_d, s.zero_w)
            table = torch.frombuffer(bytearray(b''.join(items)), dtype=torch.uint8).to(self.be.device) if items else None
            self._h2_list = (table, len(items), ptrs)
        if self.h2 and self._h2_list[1]:
            _lib.check(self.lib.ct_conv_wino_h2_pack_run(self._h2_list[0].data_ptr(), self._h2_list[1], self._s()),
                       'ct_conv_wino_h2_pack_run')

    # ------------------------------------------------------------------ forward
    def _ctx_tensors(self):
        p = {k: v.detach() for k, v in self.ctx_params.items()}
        p['scale'] = self.net._scale_value()
        return p

    def forward(self, x, use_ctx=True):
        lib, B = self.lib, self.batch
        self.generation = getattr(self, 'generation', 0) + 1      # saved activations belong to THIS forward
        self.used_ctx = bool(use_ctx and self.ctx is not None)
        if tuple(x.shape) != tuple(self.bufs['x'].shape):
            raise _lib.CtdetError('training plan was built for input %s, got %s'
                                  % (tuple(self.bufs['x'].shape), tuple(x.shape)))
        self.bufs['x'].copy_(x)
        self._repack_all()
        if self.h2:
            if self._wired_epoch != self.be.kernel_epoch:     # a step changed kernels since the slots were wired
                self._wire_absmax()
            self.be.zero_slots()                    # maxima of |activation| and |dZ| of this step (_wire_absmax)
        if self.prezero:
            self.bn_scratch.zero_()                 # one memset instead of one per BatchNorm statistics launch
        _lib.check(lib.ct_scratch_prezeroed(int(self.prezero)), 'ct_scratch_prezeroed')

        def fwd_step(st):
            if st.kind == 'pool':
                self.be.run_pool(st, self.bufs, B)
                return
            if st.kind == 'ctxpool':
                if self.used_ctx:
                    self.be.run_ctxpool(st, self.bufs, B)
                return
            if st.kind != 'conv':
                raise _lib.CtdetError('step %s has no training implementation' % st.name)
            s = self.state[st.name]
            self.be.pack_conv(s.fwd, weights=not self._batched_packs)       # epilogue vectors (bias fold)
            self.be.run_conv(s.fwd)
            if not s.is_bn:
                return
            z = self.bufs[s.zstep.dst]
            hw = st.oh * st.ow
            dst = self.bufs[st.dst]
            off = 0
            s.frozen = [not p.bn.training for p in st.parts]
            for i, p in enumerate(st.parts):
                bn = p.bn
                if s.frozen[i]:
                    # nn.BatchNorm2d in eval() mode inside a training net: running statistics, no update
                    mean_t, var_t = bn.running_mean, bn.running_var
                else:
                    mean_t, var_t = s.mean[i], s.var[i]
                    _lib.check(lib.ct_bn_train_stats(z.data_ptr(), B, z.shape[1], off, p.cout, hw,
                                                     mean_t.data_ptr(), var_t.data_ptr(), float(bn.momentum),
                                                     bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                                                     s.scratch[i].data_ptr(), self._s()),
                               st.name + ' bn stats')
                res = self.bufs[st.res] if st.res is not None else None
                _lib.check(lib.ct_bn_train_apply(
                    z.data_ptr(), mean_t.data_ptr(), var_t.data_ptr(), bn.weight.data_ptr(),
                    bn.bias.data_ptr(), float(bn.eps), int(p.relu), None,
                    res.data_ptr() if res is not None else None, res.shape[1] if res is not None else 0,
                    st.res_coff, float(st.res_scale), dst.data_ptr(), dst.shape[1], st.dst_coff + off,
                    z.shape[1], off, B, p.cout, hw, self._s()), st.name + ' bn apply')
                off += p.cout
        # Norm branch and heads on the side stream, like the inference runtime (same schedule builder)
        try:
            run_on_streams(self, fwd_step)
        finally:
            lib.ct_scratch_prezeroed(0)
        nbt = [t for t, bn in zip(self._nbt, self._bns) if bn.training]
        if nbt:
            torch._foreach_add_(nbt, 1)             # nn.BatchNorm2d's num_batches_tracked, one launch for all layers
        if self.used_ctx:
            d = self.net.num_classes
            out = self.ctx.forward(self.bufs['conf'].view(B, self.P, d), self.bufs['pool'].view(B, self.plan.M, d),
                                   self._ctx_tensors())
            return self.bufs['loc'], out, self.bufs['obj']
        return self.bufs['loc'], self.bufs['conf'], self.bufs['obj']

    # ------------------------------------------------------------------ backward
    def backward(self, dloc, dconf, dobj):
        lib, B = self.lib, self.batch
        ctx_grads = None
        if self.used_ctx:
            # Context-Transformer block: gradient of its output -> gradient of the raw conf logits
            d = self.net.num_classes
            conf3 = self.bufs['conf'].view(B, self.P, d)
            dc, dpool, ctx_grads = self.ctx.backward(conf3, self.bufs['pool'].view(B, self.plan.M, d),
                                                     self._ctx_tensors(), dconf.contiguous().view(B, self.P, -1))
            dconf = dc.view(B, -1)
            dpool = dpool.view(B, -1)
            for st in self.plan.steps:
                if st.kind == 'ctxpool':
                    ops.ctx_pool_bwd(self.bufs['conf'], st.src_base, dpool, st.dst_base, dconf, B, st.h, st.w,
                                     st.ch, st.k)
        flat = {'loc': dloc.contiguous().view(B, -1), 'conf': dconf.contiguous().view(B, -1),
                'obj': dobj.contiguous().view(B, -1)}
        written = {}

        def overlaps(name, c0, c1):
            return any(a < c1 and c0 < b for a, b in written.get(name, []))

        grads_out = [None] * len(self.params)
        main = torch.cuda.current_stream(self.be.device)
        side = self.wg_stream
        if self.prezero:
            # three memsets for the whole pass: gradient arena (bias / split weight gradients accumulate into it),
            # BatchNorm reduction scratch, Winograd weight-gradient workspaces
            self.arena.zero_()
            self.bn_scratch.zero_()
            self.wgrad_ws_all.zero_()
        if side is not None:
            side.wait_stream(main)
        join = (lambda: main.wait_stream(side)) if side is not None else None
        bk = getattr(self, 'bucketer', None)
        if bk is not None:
            bk.begin()

        def put(prm, g):
            i = self._prod_index[id(prm)]
            v = self._arena_view(prm)
            if g.data_ptr() != v.data_ptr():        # produced elsewhere (Context-Transformer block): stage it
                v.copy_(g.reshape(v.shape))
            if bk is not None:                      # all-reduce per bucket as soon as its last gradient exists;
                bk.ready(i, join)                   # the collective is ordered after BOTH streams

        if ctx_grads is not None:
            for k, prm in self.ctx_params.items():
                put(prm, ctx_grads[k])
        _lib.check(lib.ct_scratch_prezeroed(int(self.prezero)), 'ct_scratch_prezeroed')
        try:
            self._backward_steps(flat, written, overlaps, put, main, side)
        finally:
            lib.ct_scratch_prezeroed(0)
        if side is not None:
            main.wait_stream(side)
        if bk is not None:
            bk.finish()
        snap = self.arena.clone()                   # the arena is rewritten by the next backward
        for prm in self.params:
            grads_out[self._pindex[id(prm)]] = self._arena_view(prm, snap)
        return grads_out

    def _backward_steps(self, flat, written, overlaps, put, main, side):
        """Reverse walk over the plan: per fused conv the epilogue gradient, the weight gradient (side stream) and the
        data gradient."""
        lib, B = self.lib, self.batch
        for st in reversed(self.plan.steps):
            if st.kind == 'ctxpool':
                continue
            if st.kind == 'pool' and st.name in self._pool_bwd and not overlaps(st.src, 0, st.ch):
                pr = self._pool_bwd[st.name]
                sp = self.state[pr.name]
                y = self.bufs[pr.dst]
                _lib.check(lib.ct_maxpool2x2_bias_relu_bwd(y.data_ptr(), y.shape[1], pr.dst_coff, self.grads[st.dst].data_ptr(), B, st.ch,
                                                           st.h, st.w, st.oh, st.ow, sp.dz.data_ptr(), sp.zc, 0, sp.dbias[0].data_ptr(),
                                                           getattr(sp, 'dz_amax', None), self._s()), st.name + ' + ' + pr.name + ' bias bwd')
                sp.dz_from_pool = True
                continue
            if st.kind == 'pool':
                acc = overlaps(st.src, 0, st.ch)
                _lib.check(lib.ct_maxpool2d_bwd(self.bufs[st.src].data_ptr(), self.grads[st.dst].data_ptr(),
                                                self.grads[st.src].data_ptr(), B * st.ch, st.h, st.w, st.oh,
                                                st.ow, st.k, st.stride, st.pad, int(acc), self._s()), st.name)
                written.setdefault(st.src, []).append((0, st.ch))
                continue
            s = self.state[st.name]
            hw, ctot = st.oh * st.ow, s.zc          # channel stride of dZ (st.cout, padded for the heads)
            if st.segs:                                   # heads: gather the flattened gradients
                segs = (_lib.OutSegment * 3)()
                for g, sg in enumerate(st.segs):
                    t = flat[sg.dst]
                    segs[g].ptr = t.data_ptr()
                    segs[g].co_begin, segs[g].co_end, segs[g].pix_stride = sg.co_begin, sg.co_end, sg.pix_stride
                    segs[g].img_stride, segs[g].base = t.shape[1], sg.base
                _lib.check(lib.ct_head_grad_gather(segs, len(st.segs), B, ctot, hw, s.dz.data_ptr(), self._s()),
                           st.name + ' head gather')
                off = 0
                for i, p in enumerate(st.parts):
                    _lib.check(lib.ct_bias_act_backward(s.dz.data_ptr(), ctot, off, None, 0, 0, 0, B, p.cout, hw,
                                                        s.dz.data_ptr(), ctot, off, s.dbias[i].data_ptr(),
                                                        self._s()), st.name + ' bias bwd')
                    put(p.bias, s.dbias[i])
                    off += p.cout
            else:
                gy, y = self.grads[st.dst], self.bufs[st.dst]
                off = 0
                for i, p in enumerate(st.parts):
                    if s.is_bn:
                        z = self.bufs[s.zstep.dst]
                        dres, dres_ctot, dres_acc = None, 0, 0
                        if st.res is not None:
                            dr = self.grads[st.res]
                            dres, dres_ctot = dr.data_ptr(), dr.shape[1]
                            dres_acc = int(overlaps(st.res, st.res_coff, st.res_coff + p.cout))
                            written.setdefault(st.res, []).append((st.res_coff, st.res_coff + p.cout))
                        frozen = s.frozen[i]
                        mean_t, var_t = (p.bn.running_mean, p.bn.running_var) if frozen else (s.mean[i], s.var[i])
                        _lib.check((lib.ct_bn_eval_backward if frozen else lib.ct_bn_train_backward)(
                            gy.data_ptr(), gy.shape[1], st.dst_coff + off, y.data_ptr(), y.shape[1], st.dst_coff + off,
                            z.data_ptr(), mean_t.data_ptr(), var_t.data_ptr(), p.bn.weight.data_ptr(),
                            float(p.bn.eps), int(p.relu), None, float(st.res_scale), dres, dres_ctot, st.res_coff,
                            dres_acc, s.dz.data_ptr(), s.dgamma[i].data_ptr(), s.dbeta[i].data_ptr(), ctot, off,
                            B, p.cout, hw, s.scratch[i].data_ptr(), self._s()), st.name + ' bn bwd')
                        put(p.bn.weight, s.dgamma[i])
                        put(p.bn.bias, s.dbeta[i])
                    elif getattr(s, 'dz_from_pool', False):
                        s.dz_from_pool = False          # dZ and dbias came from the pool's fused backward (above)
                        if p.bias is not None:
                            put(p.bias, s.dbias[i])
                    else:
                        _lib.check(lib.ct_bias_act_backward_amax(
                            gy.data_ptr(), gy.shape[1], st.dst_coff + off, y.data_ptr(), y.shape[1], st.dst_coff + off,
                            int(p.relu), B, p.cout, hw, s.dz.data_ptr(), ctot, off,
                            s.dbias[i].data_ptr() if p.bias is not None else None, getattr(s, 'dz_amax', None), self._s()),
                            st.name + ' bias bwd')
                        if p.bias is not None:
                            put(p.bias, s.dbias[i])
                    off += p.cout
            # weight gradient of the fused conv, split back to its parts.  Nothing downstream in this backward pass
            # reads it, so it runs on the side stream next to the data-gradient chain (dz ready -> side stream).
            if side is not None:
                s.ev_dz.record(main)
                side.wait_event(s.ev_dz)
            with torch.cuda.stream(side if side is not None else main):
                if s.wgrad_wino and s.wgrad_tile == 44:
                    _lib.check(lib.ct_conv2d_wgrad_wino4s(C.byref(s.wgrad), s.dz.data_ptr(), ctot, 0, s.dw.data_ptr(),
                                                          self.wgrad_ws4s.data_ptr(), self.wgrad_ws4s.numel(), self._s()),
                               st.name + ' wgrad (winograd 4s)')
                elif s.wgrad_wino:
                    fn = lib.ct_conv2d_wgrad_wino4 if s.wgrad_tile == 4 else lib.ct_conv2d_wgrad_wino
                    _lib.check(fn(C.byref(s.wgrad), s.dz.data_ptr(), ctot, 0, s.dw.data_ptr(),
                                  (s.wgrad_ws if self.prezero else self.wgrad_ws).data_ptr(), self._s()),
                               st.name + ' wgrad (winograd)')
                else:
                    _lib.check(lib.ct_conv2d_wgrad(C.byref(s.wgrad), s.dz.data_ptr(), ctot, 0, s.dw.data_ptr(),
                                                   self._s()), st.name + ' wgrad')
            off = 0
            for p in st.parts:
                put(p.weight, s.dw[off:off + p.cout])
                off += p.cout
            # data gradient into the producer's gradient buffer (accumulate if already written)
            if s.dgrad is not None:
                acc = overlaps(st.src, st.src_coff, st.src_coff + st.cin)
                if s.dgrad_wino is not None:
                    if not self._batched_packs:
                        self._pack_dgrad(st, s)
                    s.dgrad_wino.res = self.grads[st.src].data_ptr() if acc else None
                    var4s = 3 if s.dgrad_tile == 47 else 1
                    if s.dgrad_tile in (44, 47) and st.dil > 1 and acc:
                        g = self.grads[st.src]
                        tmp = torch.empty((g.shape[0], st.cin, st.h, st.w), device=g.device)
                        w3 = _lib.ConvDesc()
                        C.memmove(C.byref(w3), C.byref(s.dgrad_wino), C.sizeof(w3))
                        w3.res, w3.out, w3.out_ctot, w3.out_coff = None, tmp.data_ptr(), st.cin, 0
                        _lib.check(lib.ct_conv2d_wino4s_fwd(C.byref(w3), s.U_d.data_ptr(), self.dgrad_ws4s.data_ptr(),
                                                            self.dgrad_ws4s.numel(), var4s, self._s()), st.name + ' dgrad (winograd 4s, dilated)')
                        g[:, st.src_coff:st.src_coff + st.cin] += tmp
                    elif s.dgrad_tile in (44, 47):
                        _lib.check(lib.ct_conv2d_wino4s_fwd(C.byref(s.dgrad_wino), s.U_d.data_ptr(), self.dgrad_ws4s.data_ptr(),
                                                            self.dgrad_ws4s.numel(), var4s, self._s()), st.name + ' dgrad (winograd 4s)')
                    elif s.dgrad_tile in (46, 48):
                        _lib.check(lib.ct_conv2d_wino4f_pool_fwd_v(C.byref(s.dgrad_wino), s.U_d.data_ptr(), 2 if s.dgrad_tile == 48 else 1,
                                                                   None, 0, 0, 0, 0, 1, self._s()), st.name + ' dgrad (winograd 4f)')
                    else:
                        run = lib.ct_conv2d_wino4_fwd if s.dgrad_tile == 4 else lib.ct_conv2d_wino_fwd
                        _lib.check(run(C.byref(s.dgrad_wino), s.U_d.data_ptr(), self._s()), st.name + ' dgrad (winograd)')
                else:
                    if not self._batched_packs:
                        self._pack_dgrad(st, s)
                    s.dgrad.res = self.grads[st.src].data_ptr() if acc else None
                    if s.dgrad_x3 is not None:
                        _lib.check(lib.ct_conv2d_x3_fwd(C.byref(s.dgrad), s.wx3_d.data_ptr(), s.dgrad_x3, self._s()),
                                   st.name + ' dgrad (bf16x3)')
                    else:
                        _lib.check(lib.ct_conv2d_fwd(C.byref(s.dgrad), self._s()), st.name + ' dgrad')
                written.setdefault(st.src, []).append((st.src_coff, st.src_coff + st.cin))


class BackboneFunction(torch.autograd.Function):
    """(x, *params) -> (loc, conf, obj) flattened head outputs, differentiable w.r.t. the parameters."""

    @staticmethod
    def forward(ctx, rt, x, use_ctx, *params):
        ctx.rt = rt
        loc, conf, obj = rt.forward(x, use_ctx)
        ctx.generation = rt.generation
        return loc.clone(), conf.clone(), obj.clone()

    @staticmethod
    def backward(ctx, dloc, dconf, dobj):
        if ctx.rt.generation != ctx.generation:
            # the activations / BatchNorm statistics this backward needs live in buffers the runtime reuses
            raise _lib.CtdetError(
                'RFBNet training runtime: backward() of forward #%d called after forward #%d overwrote its saved '
                'activations; run loss.backward() before the next model(x) of the same batch size (the engine keeps '
                'one set of activation buffers per batch size)' % (ctx.generation, ctx.rt.generation))
        grads = ctx.rt.backward(dloc, dconf, dobj)
        return (None, None, None) + tuple(grads)
