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
