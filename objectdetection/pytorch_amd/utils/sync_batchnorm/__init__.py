"""Drop-in for utils/sync_batchnorm (only what the detectors use: SynchronizedBatchNorm2d)."""
from .batchnorm import SynchronizedBatchNorm2d          # noqa: F401
