"""Hot-path subset of the reference's `data` package: the anchor configurations, the test-time input transform
(device resize), the training-time augmentation (host decisions + one device gather launch per batch), mixup and
the VOC evaluator.  Dataset classes stay the reference's."""
from .config import *  # noqa: F401,F403
from .data_augment import BaseTransform, preproc, mixup_targets, mixup_images  # noqa: F401
