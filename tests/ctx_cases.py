"""Shared by tests/test_gpu_ctx_parity.py and tools/ctx_parity.py: the Context-Transformer parity sweep and the
error budget by stage substitution (test infrastructure; uses the CPU oracle as the checker).

models/RFB_Net_vgg.py:253-271: conf [B,P,C] -> theta/phi/g -> softmax(theta phi^T) g * Wz + conf -> L2 normalise ->
cosine classifier * scale.  With the un-scaled logits theta.phi^T (|x| ~ 1e2) the softmax is near-arg-max and the
block amplifies any perturbation of its INPUT (the conf-head output) by a factor the budget below measures."""
import types

import torch
import torch.nn.functional as F

from ctdet import synth
from oracle import rfbnet_ref


def rel(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a - b).abs().max() / b.abs().max())


def qrel(a, b, q=0.9999):
    """q-quantile of |a - b| / max|b|: a tail statistic that does not hang on the single worst near-tie."""
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    d = (a - b).abs()
    k = max(1, int(round(q * d.numel())))
    return float(d.kthvalue(k).values / b.abs().max())


def build(size=300, C=60, setting='transfer'):
    from models.RFB_Net_vgg import build_net
    net = build_net(types.SimpleNamespace(method='ours', phase=2, setting=setting), size, C)
    net.load_state_dict(synth.fill_state_dict(net.state_dict()))
    net = net.eval().cuda()
    net.device = 'cuda'
    return net


def state(net, dtype=torch.float32):
    return {k: (v.detach().cpu().to(dtype) if v.is_floating_point() else v.detach().cpu())
            for k, v in net.state_dict().items()}


def subset(batch):
    """Images of a batch the oracle is evaluated on (first, last, one in the middle): the kernels are
    batch-position invariant (tests/test_gpu_harness.py), the oracle is the expensive side."""
    return sorted({0, batch // 2, batch - 1})


def sweep_case(net, size, C, setting, batch, seed, kind, sd32=None, sd64=None, extra_threads=()):
    """-> dict with e(GPU,CPU32), e(GPU,fp64), e(CPU32,fp64) of the block's output `conf` (every element of the
    oracle's image subset) and the raw loc / obj errors vs CPU32.  extra_threads: further torch thread counts at which
    the fp32 CPU reference is evaluated again (torch's CPU convolutions split their sums by thread, so the reference moves
    with the host): 'gpu_cpu32@N' / 'cpu32_fp64@N' per count."""
    sd32 = sd32 or state(net)
    sd64 = sd64 or state(net, torch.float64)
    x = synth.images(batch, size, kind, seed)
    idx = subset(batch)
    with torch.no_grad():
        got = [t.cpu()[idx] for t in net.forward_raw(x.cuda())]
        raw_gpu = net.forward_raw(x.cuda(), init=True).cpu()[idx]
        raw32 = rfbnet_ref.forward(sd32, x[idx], size, C, 2, 'ours', setting, init=True)
        w32 = rfbnet_ref.forward(sd32, x[idx], size, C, 2, 'ours', setting, raw=True)
        w64 = rfbnet_ref.forward(sd64, x[idx].double(), size, C, 2, 'ours', setting, raw=True)
        extra = {}
        base_threads = torch.get_num_threads()
        for nt in extra_threads:
            torch.set_num_threads(int(nt))
            try:
                wn = rfbnet_ref.forward(sd32, x[idx], size, C, 2, 'ours', setting, raw=True)
            finally:
                torch.set_num_threads(base_threads)
            extra['gpu_cpu32@%d' % nt] = rel(got[1], wn[1])
            extra['cpu32_fp64@%d' % nt] = rel(wn[1], w64[1])
    return {**extra, 'batch': batch, 'seed': seed, 'kind': kind, 'images': idx,
            'gpu_cpu32': rel(got[1], w32[1]), 'gpu_fp64': rel(got[1], w64[1]), 'cpu32_fp64': rel(w32[1], w64[1]),
            'q_gpu_fp64': qrel(got[1], w64[1]), 'q_cpu32_fp64': qrel(w32[1], w64[1]),
            'loc_gpu_cpu32': rel(got[0], w32[0]), 'obj_gpu_cpu32': rel(got[2], w32[2]),
            'rawconf_gpu_cpu32': rel(raw_gpu, raw32)}


def verdict(r, tol=1e-4):
    """'ok': every element of the block's output within tol of the reference's fp32 CPU arithmetic; else 'FAIL'.  north_star's
    contract, flat: no second clause (rounds 2-4 accepted cases above 1e-4 that were within 1e-4 of fp64; the shipped kernel
    policy -- engine.ctx_policy -- holds 1e-4 against the CPU path in all nine sweep cases, with the CPU reference at 8 and at
    128 threads: profiles/r06_ctx_policy.txt; round 5's: profiles/r05_ctx_policy.txt).  What makes this a narrow pass, for the record: the
    block multiplies a perturbation of its input by ~1000 (budget below), the reference's own fp32 CPU path sits 4.8..7.2e-5
    from an fp64 evaluation and moves inside that band with the host's thread count (torch's CPU convolutions split their
    sums by thread: tests/conftest.py pins 8 threads, the count tools/gen_goldens.py captured the goldens with)."""
    return 'ok' if r['gpu_cpu32'] <= tol else 'FAIL'


def pool_from_conf(conf, size, C):
    """The context pooling (:235-244) of a flat conf tensor [B,P,C] (any dtype)."""
    B = conf.shape[0]
    maps = {300: [38, 19, 10, 5, 3, 1], 512: [64, 32, 16, 8, 4, 2, 1]}[size]
    out, off = [], 0
    for hw, mb, k in zip(maps, rfbnet_ref.MBOX[size], rfbnet_ref.CTX_POOL[size]):
        n = hw * hw * mb
        c = conf[:, off:off + n].reshape(B, hw, hw, mb * C).permute(0, 3, 1, 2)
        out.append(F.max_pool2d(c, k, k, ceil_mode=True).permute(0, 2, 3, 1).reshape(B, -1, C))
        off += n
    return torch.cat(out, 1)


def block_stages(sd, conf, cp, low=()):
    """The block in fp64 with the stages named in `low` evaluated in fp32 (stage substitution):
    'proj' theta/phi/g, 'logits' theta.phi^T, 'softmax_v' softmax + aggregation, 'tail' Wz/residual/normalise/classifier."""
    def cast(t, name):
        return t.float() if name in low else t.double()

    def lin(n, t, name):
        w, b = cast(sd[n + '.weight'], name), cast(sd[n + '.bias'], name)
        t = cast(t, name)
        return (F.linear(t, w, b) + t).double()
    theta, phi, g = lin('theta', conf, 'proj'), lin('phi', cp, 'proj'), lin('g', cp, 'proj')
    s = torch.matmul(cast(theta, 'logits'), cast(phi, 'logits').transpose(1, 2)).double()
    w = F.softmax(cast(s, 'softmax_v'), dim=2)
    agg = torch.matmul(w, cast(g, 'softmax_v')).double()
    nov = cast(conf, 'tail') + cast(agg, 'tail') * cast(sd['Wz'], 'tail')
    nov = nov / nov.norm(dim=2, keepdim=True)
    return (F.linear(nov, cast(sd['OBJ_Target.weight'], 'tail')) * cast(sd['scale'], 'tail')).double()


def budget(net, size, C, batch, seed=1234, kind='randn', nimg=2):
    """Error budget of the block's output against the fp64 oracle, by substituting one stage at a time.
    Returns an ordered list of (label, error)."""
    from ctdet import ops
    sd32, sd64 = state(net), state(net, torch.float64)
    x = synth.images(batch, size, kind, seed)
    idx = subset(batch)[:nimg]
    rows = []
    with torch.no_grad():
        out_gpu = net.forward_raw(x.cuda())[1].cpu()[idx]
        conf_gpu = net.forward_raw(x.cuda(), init=True).cpu()[idx]
        rt = net.runtime(batch)
        pool_gpu = rt.bufs['pool'].view(batch, -1, C).cpu()[idx]
        heads = [st for st in rt.plan.steps if st.kind == 'conv' and st.name.startswith('head.')]
        src_gpu = [rt.bufs[st.src].cpu()[idx] for st in heads]
        conf64 = rfbnet_ref.forward(sd64, x[idx].double(), size, C, 2, 'ours', 'transfer', init=True)
        conf32 = rfbnet_ref.forward(sd32, x[idx], size, C, 2, 'ours', 'transfer', init=True)
        out32 = rfbnet_ref.forward(sd32, x[idx], size, C, 2, 'ours', 'transfer', raw=True)[1]
        cp64 = pool_from_conf(conf64, size, C)
        want = rfbnet_ref.context_block(sd64, conf64, cp64)
        rows.append(('raw conf (block input): GPU vs fp64', rel(conf_gpu, conf64)))
        rows.append(('raw conf (block input): CPU fp32 vs fp64', rel(conf32, conf64)))
        rows.append(('block output: GPU whole path vs fp64', rel(out_gpu, want)))
        rows.append(('block output: CPU fp32 whole path vs fp64', rel(out32, want)))
        rows.append(('block output: GPU whole path vs CPU fp32', rel(out_gpu, out32)))
        # upstream only: the device's conf / pooled conf through the fp64 block
        rows.append(('  upstream only  (GPU conf+pool -> fp64 block)',
                     rel(rfbnet_ref.context_block(sd64, conf_gpu.double(), pool_gpu.double()), want)))
        rows.append(('  upstream only  (CPU fp32 conf -> fp64 block)',
                     rel(rfbnet_ref.context_block(sd64, conf32.double(), pool_from_conf(conf32.double(), size, C)), want)))
        # trunk only: the device's source maps through fp64 heads and an fp64 block
        cs = []
        for i, s in enumerate(src_gpu):
            c = F.conv2d(s.double(), sd64['conf.%d.weight' % i], sd64['conf.%d.bias' % i], 1, 1)
            cs.append(c.permute(0, 2, 3, 1).reshape(len(idx), -1))
        conf_t = torch.cat(cs, 1).view(len(idx), -1, C)
        rows.append(('    trunk only   (GPU source maps -> fp64 heads -> fp64 block)',
                     rel(rfbnet_ref.context_block(sd64, conf_t, pool_from_conf(conf_t, size, C)), want)))
        rows.append(('    raw conf of that path vs fp64', rel(conf_t, conf64)))
        # block only: the fp64 conf (rounded to fp32, which is what any fp32 path is handed) through the device kernel
        c_r = conf64.float()
        p_r = pool_from_conf(c_r, size, C)
        want_r = rfbnet_ref.context_block(sd64, c_r.double(), p_r.double())
        got_k = ops.ctx_attention(c_r.cuda().contiguous(), p_r.cuda().contiguous(), net._ctx_params(), False).cpu()
        rows.append(('  block only     (fp64 conf rounded to fp32 -> GPU attention kernel)', rel(got_k, want_r)))
        rows.append(('  block only     (same input -> torch-CPU fp32 block)',
                     rel(rfbnet_ref.context_block(sd32, c_r, p_r), want_r)))
        for st in ('proj', 'logits', 'softmax_v', 'tail'):
            rows.append(('    fp32 only in stage %-9s (others fp64)' % st,
                         rel(block_stages(sd64, c_r.double(), p_r.double(), low=(st,)), want_r)))
        # conditioning of the block: relative output change per relative input perturbation
        gen = torch.Generator().manual_seed(3)
        eps = 1e-6
        pert = conf64 + torch.randn(conf64.shape, generator=gen, dtype=torch.float64) * eps * conf64.abs().max()
        amp = rel(rfbnet_ref.context_block(sd64, pert, pool_from_conf(pert, size, C)), want) / eps
        rows.append(('  amplification of a 1e-6 (of range, gaussian) input perturbation by the fp64 block', amp))
    return rows
