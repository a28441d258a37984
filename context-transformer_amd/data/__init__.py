"""Hot-path subset of the reference's `data` package: the anchor configurations, the test-time
input transform (device resize) and the VOC evaluator.  Dataset classes and training-time
augmentation stay the reference's."""
from .config import *  # noqa: F401,F403
from .data_augment import BaseTransform  # noqa: F401
