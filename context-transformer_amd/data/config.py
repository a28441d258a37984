"""Anchor configurations (data only; values of the reference's data/config.py:10-135)."""

__all__ = ['VOCroot', 'COCOroot', 'VOC_300', 'VOC_SSD_300', 'VOC_512', 'COCO_300', 'COCO_SSD_300',
           'COCO_512', 'COCO_mobile_300']

VOCroot = 'data/VOCdevkit'
COCOroot = 'data/COCO'


def _cfg(feature_maps, min_dim, steps, min_sizes, max_sizes, aspect_ratios):
    return {'feature_maps': feature_maps, 'min_dim': min_dim, 'steps': steps, 'min_sizes': min_sizes,
            'max_sizes': max_sizes, 'aspect_ratios': aspect_ratios, 'variance': [0.1, 0.2], 'clip': True}


_FM300, _ST300 = [38, 19, 10, 5, 3, 1], [8, 16, 32, 64, 100, 300]
_FM512, _ST512 = [64, 32, 16, 8, 4, 2, 1], [8, 16, 32, 64, 128, 256, 512]
_AR_RFB6 = [[2, 3], [2, 3], [2, 3], [2, 3], [2], [2]]
_AR_SSD6 = [[2], [2, 3], [2, 3], [2, 3], [2], [2]]
_AR_RFB7 = [[2, 3], [2, 3], [2, 3], [2, 3], [2, 3], [2], [2]]
_VOC_MIN, _VOC_MAX = [30, 60, 111, 162, 213, 264], [60, 111, 162, 213, 264, 315]
_COCO_MIN, _COCO_MAX = [21, 45, 99, 153, 207, 261], [45, 99, 153, 207, 261, 315]

VOC_300 = _cfg(_FM300, 300, _ST300, _VOC_MIN, _VOC_MAX, _AR_RFB6)
VOC_SSD_300 = _cfg(_FM300, 300, _ST300, _VOC_MIN, _VOC_MAX, _AR_SSD6)
VOC_512 = _cfg(_FM512, 512, _ST512, [35.84, 76.8, 153.6, 230.4, 307.2, 384.0, 460.8],
               [76.8, 153.6, 230.4, 307.2, 384.0, 460.8, 537.6], _AR_RFB7)
COCO_300 = _cfg(_FM300, 300, _ST300, _COCO_MIN, _COCO_MAX, _AR_RFB6)
COCO_SSD_300 = _cfg(_FM300, 300, _ST300, _COCO_MIN, _COCO_MAX, _AR_SSD6)
COCO_512 = _cfg(_FM512, 512, _ST512, [20.48, 51.2, 133.12, 215.04, 296.96, 378.88, 460.8],
                [51.2, 133.12, 215.04, 296.96, 378.88, 460.8, 542.72], _AR_RFB7)
COCO_mobile_300 = _cfg([19, 10, 5, 3, 2, 1], 300, [16, 32, 64, 100, 150, 300], [45, 90, 135, 180, 225, 270],
                       [90, 135, 180, 225, 270, 315], _AR_RFB6)
