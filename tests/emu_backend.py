"""TEST INFRASTRUCTURE: replay an engine Plan with torch-CPU ops.

Interprets exactly the launch descriptors the HIP backend would hand to libctdet (fused parts,
channel slices, residual wiring, head segments, context pooling) so the plan *wiring* can be
checked against the oracle without a GPU.  Never imported by the product.
"""
import torch
import torch.nn.functional as F


class EmuBackend:
    def __init__(self):
        self.device = torch.device('cpu')

    def alloc(self, shape, dtype=torch.float32):
        return torch.full(shape, float('nan'), dtype=dtype)     # NaN-poisoned: unwritten reads show up

    def prepare_conv(self, st, bufs, batch):
        st.rt['bufs'] = bufs
        st.rt['batch'] = batch

    @staticmethod
    def param_versions(st):
        return None

    def pack_conv(self, st):
        pass

    def run_conv(self, st):
        bufs = st.rt['bufs']
        x = bufs[st.src][:, st.src_coff:st.src_coff + st.cin]
        outs = []
        for p in st.parts:
            y = F.conv2d(x, p.weight.detach(), None, st.stride, (st.ph, st.pw), st.dil)
            if p.bn is not None:
                bn = p.bn
                y = F.batch_norm(y, bn.running_mean, bn.running_var, bn.weight.detach(), bn.bias.detach(),
                                 False, 0.0, bn.eps)
            elif p.bias is not None:
                y = y + p.bias.detach().view(1, -1, 1, 1)
            outs.append((y, p.relu))
        if st.res is not None:
            assert len(outs) == 1
            y, relu = outs[0]
            y = y * st.res_scale + bufs[st.res][:, st.res_coff:st.res_coff + st.cout]
            outs = [(y, relu)]
        y = torch.cat([F.relu(o) if r else o for o, r in outs], 1)
        if st.segs:
            B = y.shape[0]
            for sg in st.segs:
                part = y[:, sg.co_begin:sg.co_end].permute(0, 2, 3, 1).reshape(B, -1)
                assert sg.pix_stride == sg.co_end - sg.co_begin
                bufs[sg.dst][:, sg.base:sg.base + part.shape[1]] = part
        else:
            bufs[st.dst][:, st.dst_coff:st.dst_coff + st.cout] = y

    def run_pool(self, st, bufs, batch):
        bufs[st.dst][:] = F.max_pool2d(bufs[st.src], st.k, st.stride, st.pad, ceil_mode=st.ceil_mode)

    def run_ctxpool(self, st, bufs, batch):
        n = st.h * st.w * st.ch
        x = bufs[st.src][:, st.src_base:st.src_base + n].view(batch, st.h, st.w, st.ch).permute(0, 3, 1, 2)
        y = F.max_pool2d(x, st.k, st.k, ceil_mode=True).permute(0, 2, 3, 1).reshape(batch, -1)
        bufs[st.dst][:, st.dst_base:st.dst_base + y.shape[1]] = y


def replay_plan_autograd(plan, leaf, x, relu_masks=None, dtype=torch.float64, pool_inputs=None, batch_stats=False):
    """Differentiable replay of an inference/training Plan with eval-mode BatchNorm (TEST INFRASTRUCTURE).

    leaf: {id(parameter): tensor requiring grad} standing in for the network's parameters; buffers (running
    statistics) are read from the modules.  relu_masks(step, part_offset, cout) -> bool tensor [B,cout,oh,ow] or
    None: when given, every ReLU is replaced by multiplication with that mask -- the activation pattern of
    ANOTHER evaluation (the device's) -- so both evaluations differentiate the same linear piece of the network
    and their gradients can be compared at rounding level (a ReLU that flips under a 1e-6 perturbation of its
    input changes weight gradients on the small maps by 1e-3..1e-2, which is what a free comparison measures).
    pool_inputs(step) -> the OTHER evaluation's input tensor of a max-pool step or None: when given, the window
    maximum is taken at that evaluation's arg-max position (first maximum in row-major order, torch's rule), for the
    same reason: two values within rounding of each other may swap places.
    batch_stats=True: BatchNorm with the statistics of the batch (nn.BatchNorm2d in train mode, what train.py:222-229
    runs), differentiated through mean and variance.
    -> (loc [B,P*4], conf [B,P*C], obj [B,P*2])"""
    pieces = {'x': [(0, x.to(dtype))]}
    flat = {}

    def full(name):
        ps = sorted(pieces[name], key=lambda t: t[0])
        off = 0
        for c0, t in ps:
            assert c0 == off, (name, c0, off)
            off += t.shape[1]
        return ps[0][1] if len(ps) == 1 else torch.cat([t for _, t in ps], 1)

    def P(prm):
        return leaf[id(prm)] if id(prm) in leaf else prm.detach().to(x.device, dtype)
    for st in plan.steps:
        if st.kind == 'pool':
            src = full(st.src)
            other = pool_inputs(st) if pool_inputs is not None else None
            if other is None:
                y = F.max_pool2d(src, st.k, st.stride, st.pad, ceil_mode=st.ceil_mode)
            else:
                _, idx = F.max_pool2d(other.float(), st.k, st.stride, st.pad, ceil_mode=st.ceil_mode, return_indices=True)
                y = src.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)
            pieces[st.dst] = [(0, y)]
            continue
        if st.kind != 'conv':
            continue
        xin = full(st.src)[:, st.src_coff:st.src_coff + st.cin]
        outs, off = [], 0
        for p in st.parts:
            y = F.conv2d(xin, P(p.weight), None, st.stride, (st.ph, st.pw), st.dil)
            if p.bn is not None:
                bn = p.bn
                if batch_stats:
                    y = F.batch_norm(y, None, None, P(bn.weight), P(bn.bias), True, 0.0, bn.eps)
                else:
                    y = F.batch_norm(y, bn.running_mean.detach().to(x.device, dtype), bn.running_var.detach().to(x.device, dtype), P(bn.weight),
                                     P(bn.bias), False, 0.0, bn.eps)
            elif p.bias is not None:
                y = y + P(p.bias).view(1, -1, 1, 1)
            if st.res is not None:
                assert len(st.parts) == 1
                y = y * st.res_scale + full(st.res)[:, st.res_coff:st.res_coff + st.cout]
            if p.relu:
                m = relu_masks(st, off, p.cout) if relu_masks is not None else None
                y = F.relu(y) if m is None else y * m.to(dtype)
            outs.append(y)
            off += p.cout
        y = outs[0] if len(outs) == 1 else torch.cat(outs, 1)
        if st.segs:
            B = y.shape[0]
            for sg in st.segs:
                flat.setdefault(sg.dst, []).append((sg.base, y[:, sg.co_begin:sg.co_end].permute(0, 2, 3, 1).reshape(B, -1)))
        else:
            pieces.setdefault(st.dst, []).append((st.dst_coff, y))
    return tuple(torch.cat([t for _, t in sorted(flat[n], key=lambda t: t[0])], 1) for n in ('loc', 'conf', 'obj'))
