"""Results format + VOC evaluation against goldens produced by the reference's
data/voc_eval.py (tools/gen_goldens.py voc)."""
import os
import pickle

import numpy as np
import pytest

from ctdet import evaluate

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'voc_eval.npz'))
CLASSES = [str(c) for c in G['classes']]
IDS = [str(i) for i in G['ids']]


def _all_boxes():
    ab = [[np.empty((0, 5), np.float32) for _ in IDS]]
    for ci in range(1, len(CLASSES)):
        ab.append([G['det_c%d_i%d' % (ci, i)] for i in range(len(IDS))])
    return ab


def _gt(ci):
    gt = {}
    for iid in IDS:
        a = G['gt_%s' % iid]
        a = a[a[:, 0] == ci]
        gt[iid] = {'bbox': a[:, 1:5], 'difficult': a[:, 5].astype(bool)}
    return gt


def test_results_lines_match_reference_format():
    ab = _all_boxes()
    for ci in range(1, len(CLASSES)):
        assert evaluate.results_lines(ab[ci], IDS) == [str(x) for x in G['lines_c%d' % ci]]


@pytest.mark.parametrize('m07', [True, False])
def test_voc_eval_matches_reference(m07):
    ab = _all_boxes()
    for ci in range(1, len(CLASSES)):
        rec, prec, ap = evaluate.voc_eval_lines(evaluate.results_lines(ab[ci], IDS), _gt(ci), 0.5, m07)
        tag = 'c%d_%s' % (ci, '07' if m07 else 'area')
        assert np.array_equal(rec, G[tag + '_rec'])
        assert np.array_equal(prec, G[tag + '_prec'])
        assert ap == float(G[tag + '_ap'])
        assert 0.05 < ap < 1.0        # the fixture exercises TP, FP, duplicates and difficult boxes


def test_evaluate_detections_mean_and_missing_images():
    ab = _all_boxes()
    gts = {CLASSES[ci]: _gt(ci) for ci in range(1, len(CLASSES))}
    aps, mean = evaluate.evaluate_detections(ab, IDS, gts, CLASSES, use_07_metric=True)
    want = [float(G['c%d_07_ap' % ci]) for ci in range(1, len(CLASSES))]
    assert [aps[c] for c in CLASSES[1:]] == want
    assert mean == pytest.approx(np.mean(want), abs=0)
    # images without objects of the class may be left out of the ground truth
    sparse = {c: {k: v for k, v in g.items() if len(v['bbox'])} for c, g in gts.items()}
    assert evaluate.evaluate_detections(ab, IDS, sparse, CLASSES)[0] == aps


def test_voc_ap_known_answers():
    rec = np.array([0.2, 0.4, 0.4, 0.8, 1.0])
    prec = np.array([1.0, 1.0, 0.67, 0.8, 0.5])
    # 11-point: t<=0.4 -> 1.0 (5 points), t in .5...8 -> 0.8 (4 points), .9,1.0 -> 0.5
    assert evaluate.voc_ap(rec, prec, True) == pytest.approx((5 * 1.0 + 4 * 0.8 + 2 * 0.5) / 11)
    assert evaluate.voc_ap(rec, prec, False) == pytest.approx(0.4 * 1.0 + 0.4 * 0.8 + 0.2 * 0.5)
    assert evaluate.voc_ap(np.array([]), np.array([]), True) == 0


def test_layout_and_pickle(tmp_path):
    per_image = [[np.empty((0, 5), np.float32), np.full((2, 5), i, np.float32), np.full((1, 5), 10 + i, np.float32)]
                 for i in range(3)]
    ab = evaluate.to_reference_all_boxes(per_image)
    assert len(ab) == 3 and len(ab[0]) == 3 and ab[2][1][0, 0] == 11
    evaluate.save_detections(ab, tmp_path / 'detections.pkl')
    back = pickle.load(open(tmp_path / 'detections.pkl', 'rb'))
    assert all(np.array_equal(back[j][i], ab[j][i]) for j in range(3) for i in range(3))
    paths = evaluate.write_voc_results(ab, ['a', 'b', 'c'], ['__background__', 'x', 'y'], str(tmp_path / 'res'))
    assert open(paths['y']).read().splitlines()[1] == 'b 11.000 12.0 12.0 12.0 12.0'


def test_file_based_voc_eval_entry_point(tmp_path):
    """data/voc_eval.py protocol: XML annotations + image-set file + results files on disk."""
    from data import voc_eval as ve
    os.makedirs(tmp_path / 'Annotations')
    for iid in IDS:
        objs = ''.join('<object><name>%s</name><pose>Unspecified</pose><truncated>0</truncated><difficult>%d</difficult>'
                       '<bndbox><xmin>%d</xmin><ymin>%d</ymin><xmax>%d</xmax><ymax>%d</ymax></bndbox></object>'
                       % (CLASSES[r[0]], r[5], r[1], r[2], r[3], r[4]) for r in G['gt_%s' % iid])
        open(tmp_path / 'Annotations' / (iid + '.xml'), 'w').write('<annotation>%s</annotation>' % objs)
    open(tmp_path / 'test.txt', 'w').write('\n'.join(IDS) + '\n')
    evaluate.write_voc_results(_all_boxes(), IDS, CLASSES, str(tmp_path))
    for ci in range(1, len(CLASSES)):
        for rep in range(2):                                  # second pass reads annots.pkl
            rec, prec, ap = ve.voc_eval(str(tmp_path / 'comp4_det_test_{:s}.txt'), str(tmp_path / 'Annotations' / '{:s}.xml'),
                                        str(tmp_path / 'test.txt'), CLASSES[ci], str(tmp_path / 'cache'), 0.5, True)
            assert np.array_equal(rec, G['c%d_07_rec' % ci]) and ap == float(G['c%d_07_ap' % ci])
    assert os.path.isfile(tmp_path / 'cache' / 'annots.pkl')


def test_coco_results_and_collate(tmp_path):
    """data/coco.py:242-274 result dicts (x, y, w+1, h+1) and data/voc0712.py:429-451 collate layout."""
    import json
    import torch
    ab = [[[], []], [np.array([[10., 20., 29., 59., 0.5]], np.float32), np.empty((0, 5), np.float32)],
          [[], np.array([[1., 2., 3., 4., 0.25], [0., 0., 9., 9., 0.75]], np.float32)]]
    res = evaluate.coco_results(ab, [42, 43], [0, 18, 7])
    assert res == [{'image_id': 42, 'category_id': 18, 'bbox': [10.0, 20.0, 20.0, 40.0], 'score': 0.5},
                   {'image_id': 43, 'category_id': 7, 'bbox': [1.0, 2.0, 3.0, 3.0], 'score': 0.25},
                   {'image_id': 43, 'category_id': 7, 'bbox': [0.0, 0.0, 10.0, 10.0], 'score': 0.75}]
    path = evaluate.write_coco_results(ab, [42, 43], [0, 18, 7], str(tmp_path / 'r.json'))
    assert json.load(open(path)) == res
    batch = [(torch.zeros(3, 4, 4), np.array([[0.1, 0.1, 0.5, 0.5, 3, 1]])),
             (torch.ones(3, 4, 4), np.zeros((0, 6)))]
    imgs, targets = evaluate.detection_collate(batch)
    assert imgs.shape == (2, 3, 4, 4) and [t.shape for t in targets] == [(1, 6), (0, 6)]
    assert targets[0].dtype == torch.float32
