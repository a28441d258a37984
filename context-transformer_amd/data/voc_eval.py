"""File-based PASCAL VOC evaluation with the reference's entry points (data/voc_eval.py):
`parse_rec`, `voc_ap`, `voc_eval(detpath, annopath, imagesetfile, classname, cachedir, ...)`.
The matching itself is ctdet.evaluate.voc_eval_lines; this module only does the file handling
(XML annotations, image-set list, `annots.pkl` cache, results file)."""
import os
import pickle
import xml.etree.ElementTree as ET

import numpy as np

from ctdet.evaluate import voc_ap, voc_eval_lines  # noqa: F401  (voc_ap re-exported)


def parse_rec(filename):
    """data/voc_eval.py:13-31: objects of one annotation file."""
    objects = []
    for obj in ET.parse(filename).findall('object'):
        box = obj.find('bndbox')
        objects.append({
            'name': obj.find('name').text,
            'pose': obj.find('pose').text,
            'truncated': int(obj.find('truncated').text),
            'difficult': int(obj.find('difficult').text),
            'bbox': [int(box.find(k).text) for k in ('xmin', 'ymin', 'xmax', 'ymax')],
        })
    return objects


def load_annotations(annopath, imagenames, cachedir):
    """data/voc_eval.py:100-124: parse once, cache as `annots.pkl`."""
    os.makedirs(cachedir, exist_ok=True)
    cachefile = os.path.join(cachedir, 'annots.pkl')
    if os.path.isfile(cachefile):
        with open(cachefile, 'rb') as f:
            return pickle.load(f)
    recs = {name: parse_rec(annopath.format(name)) for name in imagenames}
    with open(cachefile, 'wb') as f:
        pickle.dump(recs, f)
    return recs


def voc_eval(detpath, annopath, imagesetfile, classname, cachedir, ovthresh=0.5, use_07_metric=False):
    """data/voc_eval.py:67-203 -> (rec, prec, ap)."""
    with open(imagesetfile, 'r') as f:
        imagenames = [x.strip() for x in f.readlines()]
    recs = load_annotations(annopath, imagenames, cachedir)
    gt = {}
    for name in imagenames:
        R = [o for o in recs[name] if o['name'] == classname]
        gt[name] = {'bbox': np.array([o['bbox'] for o in R]).reshape(-1, 4),
                    'difficult': np.array([o['difficult'] for o in R], dtype=bool)}
    with open(detpath.format(classname), 'r') as f:
        lines = f.readlines()
    return voc_eval_lines(lines, gt, ovthresh, use_07_metric)
