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
ranks with no data-path collective (inference), so scaling is weak: every rank processes its
own 32 images; value = all images / max-over-ranks time.

Extra objects on the JSON line:
  roofline      the dominant kernel (the fp32-MFMA implicit-GEMM conv instantiation that
                accumulates the most time): algorithmic FLOPs of its launches / their duration
                measured with HIP events on the launch stream inside the timed region
                (peak = 157.3 TFLOP/s dense fp32 MFMA, MI355X_MICROARCH.md).
  cpu_baseline  the CPU oracle (port of the reference path: stock torch-CPU fp32 ops + C NMS)
                on a bounded sample (BASELINE configs[0] shape, 4 images), rank 0, N=1 only.
"""
import argparse
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


def build_net(size, num_fg, phase, setting, device):
    from models.RFB_Net_vgg import build_net as bn
    from ctdet import synth
    args = types.SimpleNamespace(method='ours', phase=phase, setting=setting)
    net = bn(args, size, num_fg)
    net.load_state_dict(synth.fill_state_dict(net.state_dict()), strict=True)
    net = net.eval().to(device)
    net.device = device
    return net


def cpu_baseline(size, num_fg, images=4, reps=3):
    """Oracle (port) timed on the host cores: forward + detect + per-class NMS + top-200."""
    from ctdet import synth
    from oracle import box_ref, nms_ref, rfbnet_ref
    nms_ref.build_c()
    threads = int(os.environ.get('CTDET_CPU_THREADS', 0))
    if threads <= 0:
        try:
            import psutil
            threads = psutil.cpu_count(logical=False) or os.cpu_count() or 1     # physical cores
        except Exception:
            threads = os.cpu_count() or 1
    torch.set_num_threads(threads)
    sd = synth.fill_state_dict(rfbnet_ref.param_shapes(size, num_fg, 1))
    x = synth.images(images, size, 'randn', 1234)
    priors = box_ref.prior_box(box_ref.ANCHOR_CFGS['VOC_%d' % size])
    times = []
    with torch.no_grad():
        for r in range(reps + 1):
            t0 = time.perf_counter()
            loc, conf, obj = rfbnet_ref.forward(sd, x, size, num_fg)
            boxes, scores = box_ref.detect(loc, conf, obj, priors)
            for i in range(images):
                nms_ref.postprocess_image(boxes[i].numpy(), scores[i].numpy(), (500, 375), nms_fn=nms_ref.nms_c)
            dt = time.perf_counter() - t0
            if r > 0:
                times.append(dt)
    med = float(np.median(times))
    return {'value': round(images / med, 3), 'unit': 'images/s', 'cores': threads, 'kind': 'port',
            'sample': '%d synthetic %dx%d images (BASELINE configs[0] shape): torch-CPU fp32 forward + '
                      'Detect + per-class C NMS + top-200, median of %d runs after 1 warm-up'
                      % (images, size, size, reps)}


def pmc_traffic(event_name, workload):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/*_pmc_traffic.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs, 2*FETCH+WRITE per
    MI355X_MICROARCH.md).  Only passes recorded for THIS workload (`__workload__` entry of the file)
    count; None when no pass covers this kernel on this workload."""
    import glob
    import re
    m = re.match(r'conv_igemm_f32<(\d+)x(\d+),(\d+)x(\d+)', event_name)
    if not m and not event_name.startswith('wino_'):
        return None
    kh, kw, bm, bn = m.groups() if m else (None,) * 4
    best = None
    for path in sorted(glob.glob(os.path.join(REPO, 'profiles', '*_pmc_traffic.json'))):
        try:
            table = json.load(open(path))
        except (OSError, ValueError):
            continue
        if table.get('__workload__', {'size': 300, 'batch': 32, 'phase': 1, 'classes': 20}) != workload:
            continue
        for k, v in table.items():
            if not m:
                if k.startswith(event_name):
                    best = v['hbm_bytes']
                continue
            f = [x.strip() for x in k[k.find('<') + 1:k.find('>')].split(',')] if '<' in k else []
            if k.startswith('conv_igemm_f32') and len(f) >= 5 and f[0] == kh and f[1] == kw and f[3] == bm and f[4] == bn:
                best = v['hbm_bytes']
    return best


def conv_roofline(rt, batch, workload):
    """Per-instantiation totals from the HIP events the engine recorded around every conv launch
    of the timed region; reports the instantiation with the most accumulated time."""
    from ctdet import _lib
    lib = _lib.lib()
    agg = {}
    bf16 = getattr(rt.backend, 'load_input', None) is not None       # NHWC bf16 path (--dtype bf16, configs[4])
    peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_F32_MFMA_TFLOPS
    for st, e0, e1 in rt.event_log:
        cfg = st.rt['desc'].config
        wino = bool(st.rt.get('wino'))
        name = 'conv_bf16_nhwc' if bf16 else 'wino_f2x2_3x3_f32' if wino else 'conv_igemm_f32<%dx%d,%s>' % (
            st.kh, st.kw, lib.ct_conv_config_name(cfg - 1).decode() if cfg > 0 else 'auto')
        a = agg.setdefault(name, [0.0, 0.0, 0, 0.0])
        a[0] += e0.elapsed_time(e1) * 1e-3
        a[1] += st.flops(batch)                # ALGORITHMIC flops (direct convolution)
        a[2] += 1
        a[3] += st.flops(batch) * (16.0 / 36.0 if wino else 1.0)      # multiply-adds actually sent to the MFMA pipe
    tot_t = sum(a[0] for a in agg.values())
    tot_f = sum(a[1] for a in agg.values())
    tot_x = sum(a[3] for a in agg.values())
    name, (t, f, n, fx) = max(agg.items(), key=lambda kv: kv[1][0])
    ach = f / t / 1e12
    traffic = pmc_traffic(name, workload)
    return {
        'bound': 'mfma', 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s',
        'frac': round(ach / peak, 4), 'traffic': traffic,
        'kernel': name, 'launches': n, 'avg_launch_us': round(t / n * 1e6, 2),
        'flops_per_launch': round(f / n),
        # Winograd F(2x2,3x3) executes 16/36 of the algorithmic multiply-adds: `frac` above is algorithmic
        # flops / peak (can exceed 1), `mfma_pipe_frac` is what the matrix pipe really sustained
        'mfma_pipe_frac': round(fx / t / 1e12 / peak, 4),
        'all_conv': {'achieved': round(tot_f / tot_t / 1e12, 2),
                     'frac': round(tot_f / tot_t / 1e12 / peak, 4),
                     'mfma_pipe_frac': round(tot_x / tot_t / 1e12 / peak, 4),
                     'time_share_of_dominant': round(t / tot_t, 3),
                     # sum of the launch durations; with the two-stream schedule launches overlap, so this
                     # can exceed the wall time of a step (and every duration includes the contention)
                     'sum_launch_ms_per_step': round(tot_t / max(1, len(rt.event_log)) * len(rt.conv_steps()) * 1e3, 3),
                     'launches_per_step': len(rt.conv_steps()),
                     'streams': 2 if getattr(rt, 'side', None) is not None else 1},
    }


def log(msg):
    if os.environ.get('CTDET_BENCH_VERBOSE', '1') != '0' and int(os.environ.get('RANK', 0)) == 0:
        sys.stderr.write('[bench %7.1fs] %s\n' % (time.perf_counter() - _T0, msg))
        sys.stderr.flush()


_T0 = time.perf_counter()


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
    ap.add_argument('--batch', type=int, default=32, help='images per GPU')
    ap.add_argument('--classes', type=int, default=20)
    ap.add_argument('--phase', type=int, default=1)
    ap.add_argument('--setting', default='transfer')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'],
                    help="bf16: NHWC bf16 activations + bf16 MFMA convolutions (BASELINE configs[4]); NOT the "
                         "headline metric, which is fp32")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    a = ap.parse_args()

    from ctdet import dist as cdist
    rank, local, world = cdist.env_world()
    if world != a.gpus and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device (the product has no CPU path)')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    cdist.init('nccl')           # RCCL; only used for the timing barrier / max-over-ranks

    from ctdet import synth
    from ctdet.pipeline import DetectionPipeline
    from layers.functions import PriorBox
    import data as cfgs

    num_fg = a.classes
    T = {('transfer', 2): 20, ('incre', 2): 20}.get((a.setting, a.phase), num_fg) if a.phase == 2 else num_fg
    log('building net')
    net = build_net(a.size, num_fg, a.phase, a.setting, dev)
    net.conv_dtype = a.dtype
    log('building pipeline (plan, weight packing%s)' % (', conv autotune' if os.environ.get('CTDET_TUNE', '1') != '0' else ''))
    priors = PriorBox(getattr(cfgs, 'VOC_%d' % a.size)).forward()
    pipe = DetectionPipeline(net, priors, a.batch, T, image_wh=(500, 375))
    x = synth.images(a.batch, a.size, 'randn', 1234 + rank).to(dev)

    def sync():
        cdist.barrier(dev)

    log('warm-up')
    for _ in range(a.warmup):
        pipe.run(x)
    torch.cuda.synchronize(dev)
    log('timed region')
    if not a.no_roofline and rank == 0:
        pipe.rt.event_log = []          # HIP events around every conv launch of the timed region
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        pipe.run(x)
    sync()
    dt = time.perf_counter() - t0
    dt = cdist.max_over_ranks(dt, dev)

    log('timed region done: %.2f ms/step' % (dt / a.steps * 1e3))
    roof = None
    if pipe.rt.event_log:
        roof = conv_roofline(pipe.rt, a.batch, {'size': a.size, 'batch': a.batch, 'phase': a.phase,
                                                    'classes': a.classes})
        pipe.rt.event_log = None
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        log('cpu baseline (oracle on host cores)')
        cpu = cpu_baseline(a.size, num_fg)
        log('cpu baseline done')

    if rank == 0:
        total_images = a.batch * world * a.steps
        counts = pipe.post.out_count.sum().item()
        line = {
            'metric': 'images/sec fwd+NMS', 'value': round(total_images / dt, 2), 'unit': 'images/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(dt / a.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
            'config': {'workload': 'RFBNet-%d VGG16 inference, bs=%d per GPU, %d fg classes, phase %d%s: '
                                   'fwd + softmax/decode + per-class NMS(0.45) + top-200; name-seeded random '
                                   'weights, randn images' % (a.size, a.batch, num_fg, a.phase,
                                                              ' ' + a.setting if a.phase == 2 else ''),
                       'global_batch': a.batch * world, 'parallelism': 'dp%d (image shards, no collective)' % world,
                       'conv_gflop_per_image': round(pipe.rt.plan.conv_flops() / a.batch / 1e9, 2),
                       'detections_per_batch': int(counts), 'conv_autotuned': bool(pipe.rt.tuned)},
            'roofline': roof, 'cpu_baseline': cpu,
        }
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as tdist
        tdist.destroy_process_group()


if __name__ == '__main__':
    main()
