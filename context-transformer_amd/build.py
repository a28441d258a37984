#!/usr/bin/env python3
"""Build libctdet.so (hand-written HIP for gfx950) in-tree with hipcc.

    python context-transformer_amd/build.py [--force] [--jobs N]

Cross-compiles without a GPU.  Output: context-transformer_amd/lib/libctdet.so (git-ignored,
travels to the GPU box with the snapshot).  No torch headers are involved: the library is a
plain C ABI (include/ctdet.h).
"""
import argparse
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
OBJDIR = os.path.join(LIBDIR, 'obj')
LIB = os.path.join(LIBDIR, 'libctdet.so')
ARCH = 'gfx950'

# source -> extra flags.  Box / NMS code must keep the reference's fp32 rounding sequence.
SOURCES = {
    'ct_api.cpp': [],
    'ct_conv.hip': [],
    'ct_wino.hip': [],
    'ct_wino4.hip': [],
    'ct_wino_x3.hip': [],
    'ct_wino4s.hip': [],
    'ct_wino4f.hip': [],
    'ct_wino_wgrad.hip': [],
    'ct_wino4_wgrad.hip': [],
    'ct_conv_bf16.hip': [],
    'ct_conv_x3.hip': [],
    'ct_pool.hip': [],
    'ct_preproc.hip': ['-ffp-contract=off'],
    'ct_attn.hip': [],
    'ct_attn_bwd.hip': [],
    'ct_train.hip': [],
    'ct_box.hip': ['-ffp-contract=off'],
    'ct_nms.hip': ['-ffp-contract=off'],
    'ct_post.hip': ['-ffp-contract=off'],
    'ct_cpu_nms.cpp': ['-ffp-contract=off'],
}
COMMON = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=' + ARCH, '-I' + os.path.join(REPO, 'include'),
          '-I' + CSRC, '-Wall', '-Wno-unused-function']


def hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def _newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def _compile(src, flags, force):
    obj = os.path.join(OBJDIR, os.path.splitext(src)[0] + '.o')
    path = os.path.join(CSRC, src)
    deps = [path, os.path.join(REPO, 'include', 'ctdet.h'), os.path.join(CSRC, 'ct_common.h'),
            os.path.join(CSRC, 'ct_attn_common.h'), os.path.join(CSRC, 'ct_wino_pack.h'),
            os.path.join(CSRC, 'ct_wino4_points.h'), os.path.join(CSRC, 'ct_wino4_emit.h'), os.path.join(CSRC, 'ct_f16x2.h'), __file__]
    if force or any(_newer(d, obj) for d in deps):
        cmd = [hipcc()] + COMMON + flags + os.environ.get('CTDET_EXTRA_FLAGS', '').split() + ['-x', 'hip', '-c', path, '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed on %s:\n%s\n%s' % (src, ' '.join(cmd), r.stderr[-8000:]))
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj, True
    return obj, False


def build(force=False, jobs=None, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = {s: f for s, f in SOURCES.items() if os.path.exists(os.path.join(CSRC, s))}
    missing = set(SOURCES) - set(srcs)
    if missing:
        raise RuntimeError('missing sources: %s' % sorted(missing))
    jobs = jobs or min(len(srcs), os.cpu_count() or 4)
    with cf.ThreadPoolExecutor(jobs) as ex:
        res = list(ex.map(lambda kv: _compile(kv[0], kv[1], force), srcs.items()))
    objs = [o for o, _ in res]
    if force or any(ch for _, ch in res) or not os.path.exists(LIB):
        cmd = [hipcc(), '-shared', '-fPIC', '--offload-arch=' + ARCH, '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s' % r.stderr[-4000:])
        if verbose:
            print('built %s (%d objects recompiled)' % (LIB, sum(ch for _, ch in res)))
    elif verbose:
        print('%s is up to date' % LIB)
    return LIB


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--force', action='store_true')
    ap.add_argument('--jobs', type=int, default=None)
    a = ap.parse_args()
    build(a.force, a.jobs)
