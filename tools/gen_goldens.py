#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python tools/gen_goldens.py

Imports Ze-Yang/Context-Transformer from /root/reference (read-only; nothing is copied),
fills its modules with the build's name-seeded synthetic weights (ctdet.synth) through
``load_state_dict`` (the key set is the frozen contract), runs the hot-path functions on
seeded inputs and stores inputs-by-seed + (sampled) outputs.  The fixtures are data only.
They pin oracle/ (tests/test_oracle_golden.py); /root/reference does not exist on the GPU box.
"""
import hashlib
import importlib.util
import os
import subprocess
import sys
import tempfile
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REPO, 'context-transformer_amd'))
from ctdet import synth  # noqa: E402  (product-side synthetic data; no oracle import here)

for k in [k for k in sys.modules if k.split('.')[0] in ('models', 'layers', 'utils', 'data')]:
    del sys.modules[k]
sys.path.insert(0, REF)
from models.RFB_Net_vgg import build_net            # noqa: E402
from layers.functions import Detect, PriorBox        # noqa: E402
from layers.modules.multibox_loss_combined import MultiBoxLoss_combined  # noqa: E402
from utils import box_utils as rbu                   # noqa: E402
from utils.nms.py_cpu_nms import py_cpu_nms          # noqa: E402

spec = importlib.util.spec_from_file_location('refcfg', os.path.join(REF, 'data/config.py'))
refcfg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(refcfg)

OUT = os.path.join(REPO, 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)
MAX_SAMPLES = 4096


def sample(t):
    """-> dict(stride, vals, sum, asum, shape) of a tensor (strided flat subsample)."""
    a = t.detach().cpu().numpy().astype(np.float32).ravel()
    stride = max(1, a.size // MAX_SAMPLES)
    if stride > 1 and stride % 2 == 0:
        stride += 1                      # odd stride walks over all channel/position residues
    return dict(stride=np.int64(stride), vals=a[::stride].copy(),
                sum=np.float64(a.astype(np.float64).sum()),
                asum=np.float64(np.abs(a.astype(np.float64)).sum()),
                shape=np.array(t.shape, dtype=np.int64))


def put(store, name, t):
    for k, v in sample(t).items():
        store['%s__%s' % (name, k)] = v


def save(fname, store):
    path = os.path.join(OUT, fname)
    np.savez_compressed(path, **store)
    print('%-28s %8.1f KB  %d arrays' % (fname, os.path.getsize(path) / 1024, len(store)))


# ---------------------------------------------------------------------------
def gen_box_ops():
    st = {}
    for name in ['VOC_300', 'VOC_512', 'COCO_300', 'COCO_512', 'VOC_SSD_300', 'COCO_SSD_300', 'COCO_mobile_300']:
        p = PriorBox(getattr(refcfg, name)).forward()
        a = p.numpy()
        st['prior_%s__shape' % name] = np.array(a.shape)
        st['prior_%s__sum' % name] = np.float64(a.astype(np.float64).sum())
        st['prior_%s__sha' % name] = np.frombuffer(hashlib.sha256(a.tobytes()).digest(), dtype=np.uint8)
        st['prior_%s__rows' % name] = a[::97].copy()
    priors = PriorBox(refcfg.VOC_300).forward()
    g = torch.Generator().manual_seed(7)
    P = priors.shape[0]
    loc = torch.randn(P, 4, generator=g)
    st['decode_out'] = rbu.decode(loc, priors, [0.1, 0.2]).numpy()[::7].copy()
    st['point_form_out'] = rbu.point_form(priors).numpy()[::7].copy()
    # ground truth boxes: G=6 incl. two GTs sharing a best prior (force-match collision) and an exact prior box
    truths = torch.tensor([[0.10, 0.12, 0.45, 0.60], [0.30, 0.30, 0.80, 0.90], [0.05, 0.55, 0.25, 0.95],
                           [0.52, 0.08, 0.97, 0.44], [0.521, 0.081, 0.969, 0.441], [0.0, 0.0, 0.03, 0.03]])
    labels = torch.tensor([[3., 1.], [7., 0.6], [1., 1.], [12., 0.4], [5., 1.], [20., 1.]])
    st['match_truths'] = truths.numpy()
    st['match_labels'] = labels.numpy()
    ov = rbu.jaccard(truths, rbu.point_form(priors))
    st['jaccard_out'] = ov.numpy()[:, ::5].copy()
    st['jaccard_rowmax'] = ov.max(1)[0].numpy()
    st['jaccard_argmax0'] = ov.max(0)[1].numpy().astype(np.int64)
    matched = truths[ov.max(0)[1]]
    st['encode_out'] = rbu.encode(matched, priors, [0.1, 0.2]).numpy()[::7].copy()
    for thr in (0.5, 0.35):
        loc_t = torch.zeros(1, P, 4); conf_t = torch.zeros(1, P, 2); obj_t = torch.zeros(1, P, dtype=torch.bool)
        ovl = torch.zeros(1, P)
        rbu.match(thr, truths, priors, [0.1, 0.2], labels, loc_t, conf_t, obj_t, 0, overlap=ovl)
        tag = 'match%02d' % int(thr * 100)
        st[tag + '_loc'] = loc_t[0].numpy()[::7].copy()
        st[tag + '_conf'] = conf_t[0].numpy()
        st[tag + '_obj'] = obj_t[0].numpy()
        st[tag + '_overlap'] = ovl[0].numpy()[::7].copy()
    # Detect on random softmaxed heads, B=2, T=20
    g = torch.Generator().manual_seed(11)
    locb = torch.randn(2, P, 4, generator=g)
    conf = torch.softmax(torch.randn(2, P, 20, generator=g) * 2, -1)
    obj = torch.softmax(torch.randn(2, P, 2, generator=g), -1)
    boxes, scores = Detect(21, 0, refcfg.VOC_300).forward((locb, conf, obj), priors)
    put(st, 'detect_boxes', boxes); put(st, 'detect_scores', scores)
    # box_utils.nms (torch greedy, no +1, top_k) on decoded boxes / tie-free scores
    sc = torch.rand(P, generator=g)
    for (ovt, topk) in ((0.5, 200), (0.3, 400)):
        keep, count = rbu.nms(boxes[0], sc, ovt, topk)
        st['bunms_%02d_%d_keep' % (int(ovt * 100), topk)] = keep[:count].numpy().astype(np.int64)
    # known-answer anchors quoted in SURVEY 8c
    st['ka_decode'] = rbu.decode(torch.tensor([[1, -2, .5, -.5], [0, 0, 0, 0.]]),
                                 torch.tensor([[.5, .5, .2, .4], [.1, .9, .3, .3]]), [0.1, 0.2]).numpy()
    st['ka_jaccard'] = rbu.jaccard(torch.tensor([[0, 0, .5, .5]]),
                                   torch.tensor([[.25, .25, .75, .75], [0, 0, .5, .5], [.6, .6, .9, .9]])).numpy()
    st['ka_encode'] = rbu.encode(torch.tensor([[.3, .3, .7, .8]]), torch.tensor([[.5, .5, .2, .4]]), [0.1, 0.2]).numpy()
    save('box_ops.npz', st)


# ---------------------------------------------------------------------------
def build_patched_cpu_nms():
    """cpu_nms.pyx does not cythonize against numpy 2.x; apply the 3-token API patch to a
    COPY in a temp dir (np.int_t->np.intp_t, np.int->np.intp, np.float->float), build it
    there and import it.  Only its OUTPUTS are stored.  Returns module or None."""
    try:
        tmp = tempfile.mkdtemp(prefix='ctref_nms_')
        src = open(os.path.join(REF, 'utils/nms/cpu_nms.pyx')).read()
        src = src.replace('np.int_t', 'np.intp_t').replace('dtype=np.int)', 'dtype=np.intp)')
        src = src.replace('np.float thresh', 'float thresh')
        open(os.path.join(tmp, 'ref_cpu_nms.pyx'), 'w').write(src)
        open(os.path.join(tmp, 'setup.py'), 'w').write(
            "from setuptools import setup, Extension\nfrom Cython.Build import cythonize\nimport numpy\n"
            "setup(ext_modules=cythonize([Extension('ref_cpu_nms',['ref_cpu_nms.pyx'],"
            "include_dirs=[numpy.get_include()])], language_level=2))\n")
        subprocess.check_call([sys.executable, 'setup.py', 'build_ext', '--inplace'], cwd=tmp,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        sys.path.insert(0, tmp)
        import ref_cpu_nms
        return ref_cpu_nms
    except Exception as e:  # pragma: no cover
        print('patched cpu_nms build failed:', e)
        return None


def gen_nms():
    st = {}
    ref_ge = build_patched_cpu_nms()
    st['have_ge'] = np.int64(ref_ge is not None)
    cases = []
    rng = np.random.RandomState(2024)
    for n in (1, 2, 5, 63, 64, 65, 128, 200, 400, 1000, 2500):
        cases.append(synth.clustered_dets(n, rng=rng))
    # integer-grid boxes: many IoUs hit simple fractions exactly (stresses > vs >=)
    for n in (64, 300):
        xy = rng.randint(0, 40, (n, 2)).astype(np.float32) * 5
        wh = rng.randint(1, 8, (n, 2)).astype(np.float32) * 5 - 1
        sc = rng.permutation(np.linspace(0.02, 0.98, n)).astype(np.float32)
        cases.append(np.concatenate([xy, xy + wh, sc[:, None]], 1).astype(np.float32))
    st['ncases'] = np.int64(len(cases))
    for ci, d in enumerate(cases):
        st['c%d_dets' % ci] = d
        for thr in (0.45, 0.3, 0.5, 0.7):
            tag = 'c%d_t%02d' % (ci, int(round(thr * 100)))
            st[tag + '_gt'] = np.asarray(py_cpu_nms(d, thr), dtype=np.int64)
            if ref_ge is not None:
                st[tag + '_ge'] = np.asarray(ref_ge.cpu_nms(d, thr), dtype=np.int64)
    # known answers
    ka = np.array([[10, 10, 60, 60, .9], [12, 12, 62, 62, .8], [100, 100, 150, 150, .7], [10, 10, 60, 110, .6]], np.float32)
    st['ka_dets'] = ka
    st['ka_gt'] = np.asarray(py_cpu_nms(ka, 0.45), dtype=np.int64)
    eq = np.array([[0, 0, 9, 9, .9], [0, 0, 9, 19, .8]], np.float32)     # IoU exactly 0.5
    st['eq_dets'] = eq
    st['eq_gt'] = np.asarray(py_cpu_nms(eq, 0.5), dtype=np.int64)
    if ref_ge is not None:
        st['eq_ge'] = np.asarray(ref_ge.cpu_nms(eq, 0.5), dtype=np.int64)
        # soft-NMS (in place) for the three methods
        for m in (0, 1, 2):
            b = cases[6].copy()
            keep = ref_ge.cpu_soft_nms(b, 0.5, 0.3, 0.001, m)
            st['soft_m%d_boxes' % m] = b
            st['soft_m%d_n' % m] = np.int64(len(keep))
    save('nms.npz', st)


# ---------------------------------------------------------------------------
def make_net(size, C, phase, setting='transfer', method='ours'):
    args = types.SimpleNamespace(method=method, phase=phase, setting=setting)
    net = build_net(args, size, C)
    sd = synth.fill_state_dict(net.state_dict())
    missing = net.load_state_dict(sd, strict=True)
    net.device = 'cpu'
    return net


def hook_sources(net, store, tag):
    hs = []
    grab = {'base22': net.base[22], 'norm': net.Norm, 'conv7': net.base[34]}
    for k, m in enumerate(net.extras):
        grab['extra%d' % k] = m
    for name, m in grab.items():
        hs.append(m.register_forward_hook(lambda mod, i, o, name=name: put(store, '%s_%s' % (tag, name), o)))
    return hs


def gen_model():
    with torch.no_grad():
        # phase 1, 300, C=20, eval, B=2 -- plus intermediates
        st = {}
        net = make_net(300, 20, 1).eval()
        st['keys'] = np.array(list(net.state_dict().keys()))
        st['nparams'] = np.int64(sum(p.numel() for p in net.parameters()))
        x = synth.images(2, 300, 'randn', 1234)
        hs = hook_sources(net, st, 'p1')
        loc, conf, obj = net(x)
        for h in hs:
            h.remove()
        put(st, 'p1_loc', loc); put(st, 'p1_conf', conf); put(st, 'p1_obj', obj)
        raw = net(x, init=True)
        put(st, 'p1_init_conf', raw)
        # image-like input
        x2 = synth.images(1, 300, 'u8', 1234)
        loc, conf, obj = net(x2)
        put(st, 'p1u8_loc', loc); put(st, 'p1u8_conf', conf); put(st, 'p1u8_obj', obj)
        # Detect + test.py pipeline on the model outputs (B=2), image scale (500,375)
        priors = PriorBox(refcfg.VOC_300).forward()
        loc, conf, obj = net(x)
        boxes, scores = Detect(21, 0, refcfg.VOC_300).forward((loc, conf, obj), priors)
        put(st, 'p1_boxes', boxes); put(st, 'p1_scores', scores)
        st['p1_ncand'] = np.array([[int((scores[i, :, j] > 0.01).sum()) for j in range(21)] for i in range(2)])
        net.train()
        loc, conf, obj = net(x)
        put(st, 'p1tr_loc', loc); put(st, 'p1tr_conf', conf); put(st, 'p1tr_obj', obj)
        save('rfb300_phase1.npz', st)

        # phase 2 transfer (C=60 -> T=20) and incre (C=15 -> 15+5)
        for setting, C in (('transfer', 60), ('incre', 15)):
            st = {}
            net = make_net(300, C, 2, setting).eval()
            st['keys'] = np.array(list(net.state_dict().keys()))
            st['nparams'] = np.int64(sum(p.numel() for p in net.parameters()))
            x = synth.images(2, 300, 'randn', 1234)
            loc, conf, obj = net(x)
            put(st, 'loc', loc); put(st, 'conf', conf); put(st, 'obj', obj)
            raw = net(x, init=True)
            put(st, 'init_conf', raw)
            net.train()
            loc, conf, obj = net(x)
            put(st, 'tr_loc', loc); put(st, 'tr_conf', conf); put(st, 'tr_obj', obj)
            save('rfb300_phase2_%s.npz' % setting, st)

        # 512 phase 1 (C=20), B=1
        st = {}
        net = make_net(512, 20, 1).eval()
        st['keys'] = np.array(list(net.state_dict().keys()))
        st['nparams'] = np.int64(sum(p.numel() for p in net.parameters()))
        x = synth.images(1, 512, 'randn', 1234)
        hs = hook_sources(net, st, 'p1')
        loc, conf, obj = net(x)
        for h in hs:
            h.remove()
        put(st, 'p1_loc', loc); put(st, 'p1_conf', conf); put(st, 'p1_obj', obj)
        save('rfb512_phase1.npz', st)


# ---------------------------------------------------------------------------
def gen_loss():
    st = {}
    priors = PriorBox(refcfg.VOC_300).forward()
    for tag, (C, phase, setting, ncls) in {'p1': (20, 1, 'transfer', 21), 'p2': (60, 2, 'transfer', 21)}.items():
        net = make_net(300, C, phase, setting).train()
        x = synth.images(2, 300, 'randn', 1234)
        tg = synth.targets(2, ncls, 99)
        tg[1][0, 5] = 0.37      # one mixup weight != 1
        crit = MultiBoxLoss_combined(ncls, 0.5, True, 0, True, 3, 0.5, False)
        out = net(x)
        out = tuple(o.detach().requires_grad_(True) for o in out)
        ld = crit(out, priors, tg)
        total = sum(ld.values())
        total.backward()
        for k, v in ld.items():
            st['%s_%s' % (tag, k)] = np.float64(v.item())
        for name, o in zip(('loc', 'conf', 'obj'), out):
            put(st, '%s_in_%s' % (tag, name), o)
            put(st, '%s_grad_%s' % (tag, name), o.grad)
        for i, t in enumerate(tg):
            st['%s_target%d' % (tag, i)] = t.numpy()
    save('loss.npz', st)


# ---------------------------------------------------------------------------
def gen_pipeline():
    """test.py:130-161 with py_cpu_nms as the NMS (same '>' rule as gpu_nms), on
    trained-like synthetic heads so candidate counts are realistic and tie-free."""
    st = {}
    priors = PriorBox(refcfg.VOC_300).forward()
    P = priors.shape[0]
    g = torch.Generator().manual_seed(31)
    B, T = 2, 20
    loc = torch.randn(B, P, 4, generator=g) * 0.5
    conf = torch.softmax(torch.randn(B, P, T, generator=g) * 3.0, -1)
    obj = torch.softmax(torch.randn(B, P, 2, generator=g) * 2.0 + torch.tensor([2.5, 0.0]), -1)
    boxes, scores = Detect(T + 1, 0, refcfg.VOC_300).forward((loc, conf, obj), priors)
    scale = torch.Tensor([500, 375, 500, 375])
    for i in range(B):
        b = (boxes[i] * scale).cpu().numpy()
        s = scores[i].cpu().numpy()
        allb = [None] * (T + 1)
        for j in range(1, T + 1):
            inds = np.where(s[:, j] > 0.01)[0]
            if len(inds) == 0:
                allb[j] = np.empty([0, 5], dtype=np.float32)
                continue
            c_dets = np.hstack((b[inds], s[inds, j][:, np.newaxis])).astype(np.float32, copy=False)
            keep = py_cpu_nms(c_dets, 0.45)
            allb[j] = c_dets[keep, :]
            st['img%d_cls%d_ncand' % (i, j)] = np.int64(len(inds))
        image_scores = np.hstack([allb[j][:, -1] for j in range(1, T + 1)])
        if len(image_scores) > 200:
            th = np.sort(image_scores)[-200]
            for j in range(1, T + 1):
                k = np.where(allb[j][:, -1] >= th)[0]
                allb[j] = allb[j][k, :]
        for j in range(1, T + 1):
            st['img%d_cls%d' % (i, j)] = allb[j]
    st['seed'] = np.int64(31)
    save('pipeline.npz', st)


# ---------------------------------------------------------------------------
def gen_voc_eval():
    """Reference data/voc_eval.py on a synthetic VOC-style tree written to a temp dir (XML
    annotations, image-set file, results files in the comp4 format).  `np.bool` (removed from
    numpy >= 1.24, used at data/voc_eval.py:123) is aliased for the duration of the call."""
    spec = importlib.util.spec_from_file_location('ref_voc_eval', os.path.join(REF, 'data/voc_eval.py'))
    ve = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ve)
    if not hasattr(np, 'bool'):
        np.bool = bool
    rng = np.random.RandomState(77)
    classes = ['__background__', 'aeroplane', 'bicycle', 'bird']
    nimg = 14
    ids = ['%06d' % (i + 1) for i in range(nimg)]
    tmp = tempfile.mkdtemp(prefix='ctref_voc_')
    os.makedirs(os.path.join(tmp, 'Annotations'))
    st = {'classes': np.array(classes), 'ids': np.array(ids)}
    gts = {}
    for i, iid in enumerate(ids):
        n = rng.randint(0, 5)
        objs = []
        for k in range(n):
            x1, y1 = rng.randint(0, 300), rng.randint(0, 200)
            w, h = rng.randint(20, 180), rng.randint(20, 150)
            objs.append((classes[rng.randint(1, 4)], x1, y1, x1 + w, y1 + h, int(rng.rand() < 0.2)))
        gts[iid] = objs
        xml = ['<annotation>']
        for (c, x1, y1, x2, y2, diff) in objs:
            xml.append('<object><name>%s</name><pose>Unspecified</pose><truncated>0</truncated><difficult>%d</difficult>'
                       '<bndbox><xmin>%d</xmin><ymin>%d</ymin><xmax>%d</xmax><ymax>%d</ymax></bndbox></object>'
                       % (c, diff, x1, y1, x2, y2))
        xml.append('</annotation>')
        open(os.path.join(tmp, 'Annotations', iid + '.xml'), 'w').write('\n'.join(xml))
        st['gt_%s' % iid] = np.array([[classes.index(c), x1, y1, x2, y2, d] for (c, x1, y1, x2, y2, d) in objs],
                                     dtype=np.int64).reshape(-1, 6)
    open(os.path.join(tmp, 'test.txt'), 'w').write('\n'.join(ids) + '\n')
    # detections: jittered copies of the ground truth (some duplicates) + random false positives
    for ci in range(1, 4):
        cls = classes[ci]
        per_img = []
        for iid in ids:
            rows = []
            for (c, x1, y1, x2, y2, d) in gts[iid]:
                if c == cls:
                    for rep in range(rng.randint(1, 3)):
                        j = rng.normal(0, 6, 4)
                        rows.append([x1 + j[0], y1 + j[1], x2 + j[2], y2 + j[3], rng.uniform(0.3, 1.0)])
            for _ in range(rng.randint(0, 3)):
                x1, y1 = rng.uniform(0, 300), rng.uniform(0, 200)
                rows.append([x1, y1, x1 + rng.uniform(20, 150), y1 + rng.uniform(20, 150), rng.uniform(0.01, 0.6)])
            per_img.append(np.array(rows, dtype=np.float32).reshape(-1, 5))
        for i, a in enumerate(per_img):
            st['det_c%d_i%d' % (ci, i)] = a
        path = os.path.join(tmp, 'comp4_det_test_%s.txt' % cls)
        with open(path, 'wt') as f:
            for iid, dets in zip(ids, per_img):
                for k in range(dets.shape[0]):
                    f.write('{:s} {:.3f} {:.1f} {:.1f} {:.1f} {:.1f}\n'.format(
                        iid, dets[k, -1], dets[k, 0] + 1, dets[k, 1] + 1, dets[k, 2] + 1, dets[k, 3] + 1))
        st['lines_c%d' % ci] = np.array(open(path).read().splitlines())
        for m07 in (True, False):
            rec, prec, ap = ve.voc_eval(os.path.join(tmp, 'comp4_det_test_{:s}.txt'), os.path.join(tmp, 'Annotations', '{:s}.xml'),
                                        os.path.join(tmp, 'test.txt'), cls, os.path.join(tmp, 'cache'), 0.5, m07)
            tag = 'c%d_%s' % (ci, '07' if m07 else 'area')
            st[tag + '_rec'], st[tag + '_prec'], st[tag + '_ap'] = rec, prec, np.float64(ap)
    save('voc_eval.npz', st)


# ---------------------------------------------------------------------------
def gen_solver():
    """Reference utils/solver.py on the reference's own RFBNet: per-tensor LR groups (name order is
    what optimizer checkpoints depend on) and the warm-up / multi-step schedule."""
    spec = importlib.util.spec_from_file_location('ref_solver', os.path.join(REF, 'utils/solver.py'))
    rs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rs)
    st = {}
    for tag, (method, phase, setting, C) in {'p2ours': ('ours', 2, 'transfer', 20), 'p1': ('ours', 1, 'transfer', 60),
                                              'p2ft': ('ft', 2, 'incre', 20)}.items():
        args = types.SimpleNamespace(method=method, phase=phase, setting=setting, lr=4e-3, weight_decay=5e-4,
                                     momentum=0.9, steps=[30, 50], warmup_iter=10)
        net = build_net(args, 300, C)
        opt = rs.build_optimizer(args, net)
        names = [k for k, v in net.named_parameters() if v.requires_grad]
        st[tag + '_names'] = np.array(names)
        st[tag + '_lr'] = np.array([g['lr'] for g in opt.param_groups])
        st[tag + '_wd'] = np.array([g['weight_decay'] for g in opt.param_groups])
        st[tag + '_numel'] = np.array([g['params'][0].numel() for g in opt.param_groups])
        sched = rs.build_lr_scheduler(args, opt)
        rows = []
        for it in range(60):
            rows.append([opt.param_groups[0]['lr'], opt.param_groups[-1]['lr']])
            opt.step()
            sched.step()
        st[tag + '_sched'] = np.array(rows)
    save('solver.npz', st)


# ---------------------------------------------------------------------------
def gen_reweight():
    """train.py:252-286 (`init_reweight`) cannot be imported (argparse + cv2 at module import); its body is
    torch glue around the reference's `match`, replayed here line by line on the reference's own functions."""
    st = {}
    priors = PriorBox(refcfg.VOC_300).forward()
    P = priors.shape[0]
    for tag, (ncls, C, setting) in {'transfer': (21, 60, 'transfer'), 'incre': (21, 15, 'incre')}.items():
        g = torch.Generator().manual_seed(31 if setting == 'transfer' else 32)
        cls_list = [torch.empty(0) for _ in range(ncls - 1)]
        for it in range(2):
            num = 3
            conf_data = torch.randn(num, P, C, generator=g)
            targets = []
            for b in range(num):
                G = 7
                xy = torch.rand(G, 2, generator=g) * 0.5
                wh = torch.rand(G, 2, generator=g) * 0.4 + 0.1
                lab = ((torch.arange(G) + 7 * (it * num + b)) % (ncls - 1) + 1).float()[:, None]
                targets.append(torch.cat([xy, xy + wh, lab, torch.ones(G, 1)], 1))
            st['%s_targets_%d' % (tag, it)] = torch.stack(targets).numpy()
            loc_t = torch.Tensor(num, P, 4); conf_t = torch.Tensor(num, P, 2); obj_t = torch.BoolTensor(num, P)
            for idx in range(num):
                rbu.match(0.5, targets[idx][:, :-2], priors, [0.1, 0.2], targets[idx][:, -2:], loc_t, conf_t, obj_t, idx)
            lists = [conf_data[conf_t[:, :, 0] == i] for i in range(1, ncls)]
            cls_list = [torch.cat((cls_list[i], lists[i]), 0) for i in range(ncls - 1)]
        st[tag + '_counts'] = np.array([len(c) for c in cls_list])
        cls_list = [(item / item.norm(dim=1, keepdim=True)).mean(0) for item in cls_list]
        if setting == 'incre':
            cls_list = cls_list[15:]
        st[tag + '_weight'] = torch.stack([item / item.norm() for item in cls_list], 0).numpy()
    save('reweight.npz', st)


# ---------------------------------------------------------------------------
def gen_checkpointer():
    """The reference's utils/checkpointer.py (DetectionCheckpointer.load :259-297, _load_model :169-207, save /
    PeriodicCheckpointer :48-71,:300-349) executed on the reference's own RFBNet over the scenarios of
    tests/ckpt_cases.py.  The module imports `termcolor` (absent here) only to colour two log messages
    (:362,:380); a stand-in module whose `colored` returns its text unchanged is injected for the import."""
    stub = types.ModuleType('termcolor')
    stub.colored = lambda text, *a, **k: text
    sys.modules.setdefault('termcolor', stub)
    spec = importlib.util.spec_from_file_location('ref_checkpointer', os.path.join(REF, 'utils/checkpointer.py'))
    rc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rc)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    import ckpt_cases

    def make_model(phase):
        return build_net(types.SimpleNamespace(method='ours', phase=phase, setting='transfer'), 300, 60)
    with tempfile.TemporaryDirectory() as tmp:
        obs = ckpt_cases.run(make_model, rc, tmp)
    st = {k: np.array(v) for k, v in obs.items()}
    for k in sorted(obs):
        print('  %-16s %s' % (k, obs[k] if len(obs[k]) <= 4 else '%d entries, first %s' % (len(obs[k]), obs[k][:2])))
    save('checkpointer.npz', st)


# ---------------------------------------------------------------------------
def gen_init():
    """models/RFB_Net_vgg.py:297-318 (`init_weight`, `normalize`) and the Context-Transformer initialisation
    (:157-188): per state-dict key the statistics of what the reference's constructor leaves behind
    (std, mean, max |x|, numel, requires_grad) -- the values themselves depend on the RNG stream, the RULE
    (zeros / ones / normal with a given std / uniform with a given bound) does not -- and `normalize()` on a
    seeded classifier."""
    st = {}
    for tag, (size, C, phase, setting) in {'300_p2_transfer': (300, 60, 2, 'transfer'), '300_p2_incre': (300, 15, 2, 'incre'),
                                           '512_p1': (512, 20, 1, 'transfer')}.items():
        torch.manual_seed(11)
        net = build_net(types.SimpleNamespace(method='ours', phase=phase, setting=setting), size, C)
        grads = dict((k, v.requires_grad) for k, v in net.named_parameters())
        keys, rows = [], []
        for k, v in net.state_dict().items():
            if k == 'Wz':
                v = torch.zeros_like(v)          # torch.FloatTensor(60).fill_(0) at :171/:188
            a = v.double()
            keys.append(k)
            rows.append([float(a.std()) if a.numel() > 1 else 0.0, float(a.mean()), float(a.abs().max()), a.numel(),
                         float(grads.get(k, False))])
        st[tag + '_keys'] = np.array(keys)
        st[tag + '_stats'] = np.array(rows)
        if phase == 2:
            g = torch.Generator().manual_seed(77)
            w = torch.randn(net.OBJ_Target.weight.shape, generator=g)
            net.OBJ_Target.weight.data = w.clone()
            net.normalize()
            st[tag + '_normalized'] = net.OBJ_Target.weight.data.numpy().copy()
    save('init.npz', st)


# ---------------------------------------------------------------------------
def gen_augment():
    """data/data_augment.py:164-221 (`preproc.__call__`) executed for its TARGET path and its random-number
    consumption.  The module needs cv2 (absent) for pixels only: a stand-in whose `cvtColor` returns its input and
    whose `resize` returns a blank image lets the reference's own `_crop` / `_distort` / `_expand` / `_mirror` /
    box arithmetic run and draw exactly the random numbers it always draws; the images it returns are discarded.
    Stored per case: the input targets, image shape, seed, the returned targets and a digest of the generator state
    after the call (pins the NUMBER and order of draws, i.e. every decision branch)."""
    import random
    cv2 = types.ModuleType('cv2')
    for i, n in enumerate(('INTER_LINEAR', 'INTER_CUBIC', 'INTER_AREA', 'INTER_NEAREST', 'INTER_LANCZOS4',
                           'COLOR_BGR2HSV', 'COLOR_HSV2BGR')):
        setattr(cv2, n, i)
    cv2.cvtColor = lambda img, code: img
    cv2.resize = lambda img, size, interpolation=None: np.zeros((size[1], size[0], 3), dtype=np.uint8)
    sys.modules.setdefault('cv2', cv2)
    spec = importlib.util.spec_from_file_location('ref_data_augment', os.path.join(REF, 'data/data_augment.py'))
    da = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(da)
    st = {}
    rng = np.random.RandomState(5)
    ncase = 48
    for k in range(ncase):
        h, w = int(rng.randint(120, 400)), int(rng.randint(120, 500))
        G = int(rng.randint(1, 6))
        xy = rng.uniform(0, 0.6, (G, 2)) * (w, h)
        wh = rng.uniform(0.08, 0.4, (G, 2)) * (w, h)
        lab = rng.randint(0, 20, (G, 1)).astype(np.float64)
        tg = np.hstack([xy, np.minimum(xy + wh, (w - 1, h - 1)), lab])
        cls = None if k % 4 else int(lab[0, 0])
        random.seed(1000 + k)
        pre = da.preproc(300, (104, 117, 123), 0.6)
        img = np.zeros((h, w, 3), dtype=np.uint8)
        _, tout = pre(img, tg.copy(), cls) if cls is not None else pre(img, tg.copy())
        st['case%d_in' % k] = tg
        st['case%d_shape' % k] = np.array([h, w, -1 if cls is None else cls])
        st['case%d_out' % k] = np.asarray(tout, dtype=np.float64)
        st['case%d_state' % k] = np.frombuffer(hashlib.sha256(repr(random.getstate()).encode()).digest()[:8], dtype=np.uint8)
    st['ncase'] = np.array(ncase)
    save('augment.npz', st)


if __name__ == '__main__':
    which = sys.argv[1:] or ['init', 'augment', 'box', 'nms', 'model', 'loss', 'pipeline', 'voc', 'solver', 'reweight', 'checkpointer']
    if 'checkpointer' in which:
        gen_checkpointer()
    if 'init' in which:
        gen_init()
    if 'augment' in which:
        gen_augment()
    if 'box' in which:
        gen_box_ops()
    if 'nms' in which:
        gen_nms()
    if 'model' in which:
        gen_model()
    if 'loss' in which:
        gen_loss()
    if 'pipeline' in which:
        gen_pipeline()
    if 'voc' in which:
        gen_voc_eval()
    if 'solver' in which:
        gen_solver()
    if 'reweight' in which:
        gen_reweight()
