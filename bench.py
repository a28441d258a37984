#!/usr/bin/env python3
"""bench.py -- images/sec of forward + NMS for RFBNet-300 (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]            (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path over one synthetic batch already resident in HBM:
RFBNet engine (fused HIP convs) -> fused softmax/decode/score fusion -> per (image, class)
threshold + sort + NMS(0.45) -> per-image top-200 (test.py:130-161 generalised to a batch).
Workload = BASELINE.json configs[1]: RFBNet-300 VGG16, bs=32 per GPU, fp32, 20 foreground
classes, name-seeded random weights (no checkpoints/datasets offline).  Images shard across
ranks with no data-path collective (inference).  `--scaling weak` (default): every rank owns
--batch images; `--scaling strong`: ONE batch of --batch images is split over the ranks, the
reference's DataParallel scatter (train.py:296-297).  value = all images / max-over-ranks time.

Extra objects on the JSON line (N=1, rank 0):
  roofline      dominant kernel = the conv instantiation with the most accumulated time, from HIP
                events recorded on the launch stream around every conv launch INSIDE the timed region.
                `achieved`/`frac` count the multiply-adds the kernel's algorithm executes on the matrix
                pipe (Winograd F(2x2,3x3): 16/36 of the direct-convolution count) against the dense fp32
                MFMA peak; `algorithmic_*` is the direct-convolution FLOP count of SURVEY 8(d) over the
                same time (can exceed the peak -- that is the Winograd saving, not a roofline fraction).
                `traffic` = HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/).
    .stages     context attention / score fusion / select+sort / NMS kernels: per-launch HIP events
                recorded by the library (ct_profile_enable) in a few extra steps after the timed region,
                against the roofline that bounds each, `traffic` from the same PMC passes.
  cpu_baseline  the CPU oracle (port of the reference path: stock torch-CPU fp32 ops + C NMS) on a bounded
                sample (BASELINE configs[0] shape), all physical cores and one thread, split by stage.
  other_configs the other single-GPU configurations BASELINE.json names (512, +Context-Transformer, the
                bs-4 shard of a strong-scaled batch), a few steps each in the same process.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, 'context-transformer_amd'))
sys.path.insert(0, REPO)

PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0      # dense bf16 MFMA (MI355X_MICROARCH.md), --dtype bf16 only
PEAK_HBM_GBS = 8000.0               # HBM3E spec (6.3 TB/s achievable, MI355X_MICROARCH.md)
# multiplications executed / direct-convolution multiplications: F(2x2,3x3) 16 per 4 outputs x 9, F(4x4,3x3) 36 per 16 x 9
WINOGRAD_MULT_RATIO = {2: 16.0 / 36.0, 4: 36.0 / 144.0}
WINOGRAD_KERNEL = {2: 'wino_f2x2_3x3_f32', 4: 'wino_f4x4_3x3_f32'}


def build_net(size, num_fg, phase, setting, device):
    from models.RFB_Net_vgg import build_net as bn
    from ctdet import synth
    args = types.SimpleNamespace(method='ours', phase=phase, setting=setting)
    net = bn(args, size, num_fg)
    net.load_state_dict(synth.fill_state_dict(net.state_dict()), strict=True)
    net = net.eval().to(device)
    net.device = device
    return net


def physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n), 'psutil physical cores'
    except Exception:
        pass
    return int(os.cpu_count() or 1), 'os.cpu_count() logical CPUs'


def cpu_baseline(size, num_fg, images=4, reps=3):
    """Oracle (port) timed on the host cores, SURVEY 8(d): forward / Detect / per-class NMS + top-200 timed
    separately and end to end, with all physical cores and with one thread."""
    from ctdet import synth
    from oracle import box_ref, nms_ref, rfbnet_ref
    nms_ref.build_c()
    threads = int(os.environ.get('CTDET_CPU_THREADS', 0))
    src = 'CTDET_CPU_THREADS'
    if threads <= 0:
        threads, src = physical_cores()
    sd = synth.fill_state_dict(rfbnet_ref.param_shapes(size, num_fg, 1))
    priors = box_ref.prior_box(box_ref.ANCHOR_CFGS['VOC_%d' % size])

    def run(n_img, n_threads, n_reps):
        torch.set_num_threads(n_threads)
        x = synth.images(n_img, size, 'randn', 1234)
        rows = []
        with torch.no_grad():
            for r in range(n_reps + 1):
                t0 = time.perf_counter()
                loc, conf, obj = rfbnet_ref.forward(sd, x, size, num_fg)
                t1 = time.perf_counter()
                boxes, scores = box_ref.detect(loc, conf, obj, priors)
                t2 = time.perf_counter()
                for i in range(n_img):
                    nms_ref.postprocess_image(boxes[i].numpy(), scores[i].numpy(), (500, 375), nms_fn=nms_ref.nms_c)
                t3 = time.perf_counter()
                if r > 0:
                    rows.append((t3 - t0, t1 - t0, t2 - t1, t3 - t2))
        med = np.median(np.array(rows), axis=0)
        return {'images_per_s': round(n_img / float(med[0]), 3),
                'ms_per_image': {'forward': round(float(med[1]) / n_img * 1e3, 2),
                                 'detect': round(float(med[2]) / n_img * 1e3, 2),
                                 'nms_top200': round(float(med[3]) / n_img * 1e3, 2)}}
    full = run(images, threads, reps)
    one = run(2, 1, 1)          # bounded: two images, one timed run after one warm-up
    torch.set_num_threads(threads)
    return {'value': full['images_per_s'], 'unit': 'images/s', 'cores': threads, 'cores_source': src, 'kind': 'port',
            'stages_ms_per_image': full['ms_per_image'],
            'one_thread': {'value': one['images_per_s'], 'unit': 'images/s', 'cores': 1,
                           'stages_ms_per_image': one['ms_per_image'], 'sample': '2 images, 1 run after 1 warm-up'},
            'sample': '%d synthetic %dx%d images (BASELINE configs[0] shape): torch-CPU fp32 forward + '
                      'Detect + per-class C NMS + top-200, median of %d runs after 1 warm-up'
                      % (images, size, size, reps)}


def load_pmc(workload):
    """{kernel name prefix: HBM bytes per launch} from the newest committed rocprofv3 PMC passes recorded for THIS
    workload (profiles/*_pmc_traffic*.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs, 2*FETCH+WRITE per
    MI355X_MICROARCH.md)."""
    import glob
    best = {}
    for path in sorted(glob.glob(os.path.join(REPO, 'profiles', '*_pmc_traffic*.json'))):
        try:
            table = json.load(open(path))
        except (OSError, ValueError):
            continue
        if table.get('__workload__', {'size': 300, 'batch': 32, 'phase': 1, 'classes': 20}) != workload:
            continue
        for k, v in table.items():
            if k != '__workload__':
                best[k] = (v['hbm_bytes'], os.path.basename(path))
    return best


def pmc_lookup(pmc, name):
    """HBM bytes per launch of kernel `name` (bench naming) in the PMC table (rocprof naming)."""
    import re
    m = re.match(r'conv_igemm_f32<(\d+)x(\d+),(\d+)x(\d+)', name)
    for k, (v, _) in pmc.items():
        if m:
            f = [x.strip() for x in k[k.find('<') + 1:k.find('>')].split(',')] if '<' in k else []
            if k.startswith('conv_igemm_f32') and len(f) >= 5 and (f[0], f[1], f[3], f[4]) == m.groups():
                return v
        elif k.startswith(name):
            return v
    return None


def conv_roofline(rt, batch, pmc):
    """Per-instantiation totals from the HIP events the engine recorded around every conv launch
    of the timed region; reports the instantiation with the most accumulated time."""
    from ctdet import _lib
    lib = _lib.lib()
    agg = {}
    bf16 = getattr(rt.backend, 'load_input', None) is not None       # NHWC bf16 path (--dtype bf16, configs[4])
    peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_F32_MFMA_TFLOPS
    for st, e0, e1 in rt.event_log:
        cfg = st.rt['desc'].config
        wino = int(st.rt.get('wino') or 0)              # 0, 2 = F(2x2,3x3), 4 = F(4x4,3x3)
        name = 'conv_bf16_nhwc' if bf16 else st.rt.get('kernel_name') or (WINOGRAD_KERNEL[wino] if wino else
               'conv_igemm_f32<%dx%d,%s>' % (st.kh, st.kw, lib.ct_conv_config_name(cfg - 1).decode() if cfg > 0 else 'auto'))
        a = agg.setdefault(name, [0.0, 0.0, 0, 0.0])
        a[0] += e0.elapsed_time(e1) * 1e-3
        a[1] += st.flops(batch)                # direct-convolution flops
        a[2] += 1
        a[3] += st.flops(batch) * (WINOGRAD_MULT_RATIO[wino] if wino else 1.0)     # multiply-adds sent to the MFMA pipe
    tot_t = sum(a[0] for a in agg.values())
    tot_f = sum(a[1] for a in agg.values())
    tot_x = sum(a[3] for a in agg.values())
    name, (t, f, n, fx) = max(agg.items(), key=lambda kv: kv[1][0])
    wino = {v: k for k, v in WINOGRAD_KERNEL.items()}.get(name, 0)
    ach = fx / t / 1e12
    return {
        'bound': 'mfma', 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s',
        'frac': round(ach / peak, 4), 'traffic': pmc_lookup(pmc, name),
        'kernel': name, 'launches': n, 'avg_launch_us': round(t / n * 1e6, 2),
        'flops_per_launch': round(fx / n),
        'flops_definition': 'multiply-adds x2 executed on the matrix pipe per launch' +
                            (' = direct-convolution flops x %s (Winograd F(%dx%d,3x3), output-tile padding not counted)'
                             % ({2: '16/36', 4: '36/144'}[wino], wino, wino) if wino else ''),
        'algorithmic_flops_per_launch': round(f / n),
        'algorithmic_achieved': round(f / t / 1e12, 2),
        'algorithmic_frac': round(f / t / 1e12 / peak, 4),
        'winograd_mult_ratio': round(WINOGRAD_MULT_RATIO[wino], 4) if wino else 1.0,
        'by_kernel': {k: {'launches_per_step': round(v[2] / max(1, len(rt.event_log)) * len(rt.conv_steps()), 1),
                          'ms_per_step': round(v[0] / max(1, len(rt.event_log)) * len(rt.conv_steps()) * 1e3, 3),
                          'executed_frac': round(v[3] / v[0] / 1e12 / peak, 4),
                          'algorithmic_frac': round(v[1] / v[0] / 1e12 / peak, 4)}
                      for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:4]},
        'all_conv': {'achieved': round(tot_x / tot_t / 1e12, 2),
                     'frac': round(tot_x / tot_t / 1e12 / peak, 4),
                     'algorithmic_achieved': round(tot_f / tot_t / 1e12, 2),
                     'algorithmic_frac': round(tot_f / tot_t / 1e12 / peak, 4),
                     'time_share_of_dominant': round(t / tot_t, 3),
                     # sum of the launch durations; with the two-stream schedule launches overlap, so this
                     # can exceed the wall time of a step (and every duration includes the contention)
                     'sum_launch_ms_per_step': round(tot_t / max(1, len(rt.event_log)) * len(rt.conv_steps()) * 1e3, 3),
                     'launches_per_step': len(rt.conv_steps()),
                     'streams': 2 if getattr(rt, 'side', None) is not None else 1},
    }


def stage_rooflines(pipe, x, steps, pmc):
    """Per-launch HIP events from the library's own profile scopes (ct_profile_enable) over `steps` extra steps."""
    from ctdet import _lib
    lib = _lib.lib()
    B, P, T = pipe.batch, pipe.P, pipe.T
    torch.cuda.synchronize()
    graph_mode, pipe.use_graph = pipe.use_graph, False      # per-launch events need eager launches
    _lib.check(lib.ct_profile_enable(1), 'ct_profile_enable')
    for _ in range(steps):
        pipe.run(x)
    torch.cuda.synchronize()
    pipe.use_graph = graph_mode
    n = C.c_int(0)
    _lib.check(lib.ct_profile_collect(None, 0, C.byref(n)), 'ct_profile_collect')
    recs = (_lib.ProfileRecord * max(n.value, 1))()
    _lib.check(lib.ct_profile_collect(recs, n.value, C.byref(n)), 'ct_profile_collect')
    agg = {}
    for i in range(n.value):
        a = agg.setdefault(recs[i].name.decode(), [0.0, 0])
        a[0] += recs[i].ms * 1e-3
        a[1] += 1
    _lib.check(lib.ct_profile_enable(0), 'ct_profile_enable')
    cand = int((pipe.scores[:, :, 1:] > pipe.conf_thresh).sum().item())      # candidates over all (image, class)
    net = pipe.net
    work = {
        # loc 16P + conf 4CP + obj 8P in, boxes 16P + scores 4(T+1)P out per image, priors 16P once
        'detect_kernel': ('hbm', B * P * (16 + 4 * pipe.scores_in_ch + 8 + 16 + 4 * (T + 1)) + 16 * P),
        # threshold scan of every score + 8 B key/index and 20 B row per candidate written for the sort
        'select_sort_kernel': ('hbm', B * P * (T + 1) * 4 + cand * 28),
        # 20 B row in + 4 B kept index out per candidate (SURVEY 8d "20N in + 4N out"), both NMS passes share it
        'nms_segments_kernel': ('hbm', cand * 24),
    }
    if net.method == 'ours' and net.phase == 2:
        d, M = net.num_classes, pipe.rt.plan.M
        work['ctx_attn_kernel'] = ('mfma', B * 4.0 * P * M * d)      # QK^T and PV contractions, 2 flop per multiply-add
        # read conf P*d*4 + pooled conf M*d*4, write P*T*4 per image (4.17 MB at 300 / transfer)
        work['ctx_attn_kernel.hbm'] = ('hbm', B * (P * d * 4 + M * d * 4 + P * pipe.scores_in_ch * 4))
    out = {}
    for key, (bound, amount) in work.items():
        kname = key.split('.')[0]
        if kname not in agg:
            continue
        t, cnt = agg[kname]
        per_launch = amount / (cnt / steps)
        avg = t / cnt
        if bound == 'mfma':
            ach, peak, unit = per_launch / avg / 1e12, PEAK_F32_MFMA_TFLOPS, 'TFLOP/s'
        else:
            ach, peak, unit = per_launch / avg / 1e9, PEAK_HBM_GBS, 'GB/s'
        out[key] = {'bound': bound, 'achieved': round(ach, 2), 'peak': peak, 'unit': unit, 'frac': round(ach / peak, 4),
                    'launches_per_step': round(cnt / steps, 2), 'avg_launch_us': round(avg * 1e6, 2),
                    'algorithmic_per_launch': int(per_launch), 'traffic': pmc_lookup(pmc, kname)}
    out['other_kernels_us_per_step'] = {k: round(v[0] / steps * 1e6, 2) for k, v in sorted(agg.items())
                                        if k not in [w.split('.')[0] for w in work]}
    out['candidates_per_step'] = cand
    return out


def log(msg):
    if os.environ.get('CTDET_BENCH_VERBOSE', '1') != '0' and int(os.environ.get('RANK', 0)) == 0:
        sys.stderr.write('[bench %7.1fs] %s\n' % (time.perf_counter() - _T0, msg))
        sys.stderr.flush()


_T0 = time.perf_counter()


def make_pipeline(size, num_fg, phase, setting, batch, dtype, dev):
    from ctdet.pipeline import DetectionPipeline
    from layers.functions import PriorBox
    import data as cfgs
    T = 20 if phase == 2 else num_fg
    net = build_net(size, num_fg, phase, setting, dev)
    net.conv_dtype = dtype
    priors = PriorBox(getattr(cfgs, 'VOC_%d' % size)).forward()
    pipe = DetectionPipeline(net, priors, batch, T, image_wh=(500, 375))
    pipe.scores_in_ch = T            # channels of the conf tensor the score-fusion kernel reads
    return pipe


def quick_config(size, num_fg, phase, setting, batch, dtype, dev, steps=5, warmup=4):
    """ms/step and images/s of another configuration, same step definition, in this process."""
    from ctdet import synth
    pipe = make_pipeline(size, num_fg, phase, setting, batch, dtype, dev)
    x = synth.images(batch, size, 'randn', 4321).to(dev)
    for _ in range(warmup):
        pipe.run(x)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.run(x)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    flops = pipe.rt.plan.conv_flops()
    res = {'ms_per_step': round(dt * 1e3, 3), 'images_per_s': round(batch / dt, 1), 'batch': batch, 'steps': steps,
           'conv_gflop_per_image': round(flops / batch / 1e9, 2),
           'conv_algorithmic_tflops': round(flops / dt / 1e12, 1)}
    del pipe, x
    torch.cuda.empty_cache()
    return res


def main():
    import faulthandler
    faulthandler.enable()
    if os.environ.get('CTDET_BENCH_WATCHDOG'):
        faulthandler.dump_traceback_later(int(os.environ['CTDET_BENCH_WATCHDOG']), exit=False)
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--size', type=int, default=300)
    ap.add_argument('--batch', type=int, default=32, help='images per GPU (weak scaling) / per job (strong scaling)')
    ap.add_argument('--classes', type=int, default=20)
    ap.add_argument('--phase', type=int, default=1)
    ap.add_argument('--setting', default='transfer')
    ap.add_argument('--scaling', default=os.environ.get('CTDET_BENCH_SCALING', 'weak'), choices=['weak', 'strong'],
                    help='strong: ONE batch of --batch images split over the ranks (train.py:296-297 DataParallel scatter)')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'],
                    help="bf16: NHWC bf16 activations + bf16 MFMA convolutions (BASELINE configs[4]); NOT the "
                         "headline metric, which is fp32")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true')
    a = ap.parse_args()

    from ctdet import dist as cdist
    rank, local, world = cdist.env_world()
    if world != a.gpus and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device (the product has no CPU path)')
    ndev = torch.cuda.device_count()
    if local >= ndev:            # more ranks than GPUs (a one-GPU box rehearsing the multi-rank path): share devices
        local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    # RCCL (backend 'nccl'); only used for the timing barrier / max-over-ranks.  RCCL refuses two ranks on one device,
    # so a rehearsal with more ranks than GPUs (or CTDET_DIST_BACKEND=gloo) uses gloo for those two scalars.
    cdist.init(os.environ.get('CTDET_DIST_BACKEND') or ('nccl' if world <= ndev else 'gloo'))

    from ctdet import synth
    num_fg = a.classes
    if a.scaling == 'strong':
        b0, b1 = cdist.shard(a.batch, rank, world)       # contiguous image shard of the one global batch
        batch, global_batch = b1 - b0, a.batch
        if batch == 0:
            raise SystemExit('--scaling strong: batch %d cannot be split over %d ranks' % (a.batch, world))
    else:
        batch, global_batch = a.batch, a.batch * world
    log('building net + pipeline (plan, weight packing%s)' % (', conv autotune' if os.environ.get('CTDET_TUNE', '1') != '0' else ''))
    pipe = make_pipeline(a.size, num_fg, a.phase, a.setting, batch, a.dtype, dev)
    x = synth.images(global_batch if a.scaling == 'strong' else batch, a.size, 'randn', 1234 + (0 if a.scaling == 'strong' else rank))
    if a.scaling == 'strong':
        x = x[b0:b1]
    x = x.to(dev)

    def sync():
        cdist.barrier(dev)

    log('warm-up')
    for _ in range(a.warmup):
        pipe.run(x)
    torch.cuda.synchronize(dev)
    log('timed region')
    want_roof = not a.no_roofline and rank == 0 and world == 1      # N > 1 runs only report throughput
    graph_mode = bool(pipe.use_graph)
    if want_roof and not graph_mode:
        pipe.rt.event_log = []          # HIP events around every conv launch of the timed region
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        pipe.run(x)
    sync()
    dt = time.perf_counter() - t0
    dt = cdist.max_over_ranks(dt, dev)

    log('timed region done: %.2f ms/step' % (dt / a.steps * 1e3))
    graph_mode = pipe._graph is not None            # False if the capture failed and the steps were launched eagerly
    workload = {'size': a.size, 'batch': batch, 'phase': a.phase, 'classes': a.classes}
    roof = None
    events_from = 'the timed region'
    if want_roof and graph_mode:
        # the timed region replays a hipGraph, inside which no per-launch events can be recorded: the same K steps
        # again as eager launches with the events around every conv launch
        log('event pass (eager launches of the same steps)')
        pipe.rt.event_log = []
        for _ in range(a.steps):
            pipe.run(x)
        torch.cuda.synchronize(dev)
        events_from = 'an eager pass of the same %d steps right after the timed region (the timed region replays a ' \
                      'hipGraph)' % a.steps
    if pipe.rt.event_log:
        pmc = load_pmc(workload)
        roof = conv_roofline(pipe.rt, batch, pmc)
        roof['events_from'] = events_from
        pipe.rt.event_log = None
        log('stage rooflines (library profile scopes, 5 extra steps)')
        roof['stages'] = stage_rooflines(pipe, x, 5, pmc)
        roof['traffic_source'] = sorted({v[1] for v in pmc.values()}) or None
    counts = int(pipe.post.out_count.sum().item())
    conv_gflop = round(pipe.rt.plan.conv_flops() / batch / 1e9, 2)
    tuned = bool(pipe.rt.tuned)
    other = None
    if rank == 0 and world == 1 and not a.no_other_configs and a.dtype == 'f32' and \
            (a.size, a.phase, a.classes, a.batch) == (300, 1, 20, 32):
        del pipe
        torch.cuda.empty_cache()
        other = {}
        for key, cfg in (('rfb512_bs32', (512, 20, 1, 'transfer', 32)),
                         ('rfb300_ctx_bs32', (300, 60, 2, 'transfer', 32)),
                         ('rfb512_ctx_bs32', (512, 60, 2, 'transfer', 32)),
                         ('rfb300_bs4_strong_shard', (300, 20, 1, 'transfer', 4))):
            log('other config %s' % key)
            other[key] = quick_config(*cfg, 'f32', dev, steps=20 if cfg[4] == 4 else 5)
        other['rfb300_bs4_strong_shard']['note'] = \
            'the per-GPU shard when ONE bs-32 batch is split over 8 GPUs (--scaling strong); no collective on the path'
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        log('cpu baseline (oracle on host cores)')
        cpu = cpu_baseline(a.size, num_fg)
        log('cpu baseline done')

    if rank == 0:
        total_images = global_batch * a.steps
        line = {
            'metric': 'images/sec fwd+NMS', 'value': round(total_images / dt, 2), 'unit': 'images/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(dt / a.steps * 1e3, 3), 'higher_is_better': True, 'scaling': a.scaling,
            'launch_mode': 'hipGraph replay' if graph_mode else 'eager launches',
            'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
            'config': {'workload': 'RFBNet-%d VGG16 inference, bs=%d per GPU, %d fg classes, phase %d%s: '
                                   'fwd + softmax/decode + per-class NMS(0.45) + top-200; name-seeded random '
                                   'weights, randn images' % (a.size, batch, num_fg, a.phase,
                                                              ' ' + a.setting if a.phase == 2 else ''),
                       'global_batch': global_batch,
                       'parallelism': 'dp%d (image shards, no collective, %s scaling)' % (world, a.scaling),
                       'conv_gflop_per_image': conv_gflop,
                       'detections_per_batch': counts, 'conv_autotuned': tuned},
            'roofline': roof, 'cpu_baseline': cpu, 'other_configs': other,
        }
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as tdist
        tdist.destroy_process_group()


if __name__ == '__main__':
    main()
