"""Hot-path subset of the reference's `data` package: the anchor configurations only.
Datasets, augmentation and evaluators are out of scope."""
from .config import *  # noqa: F401,F403
