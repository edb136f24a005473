"""ksevendet/architecture/losses.py is the same `FocalLoss` / `calc_iou` as architectures/losses.py."""
from ...architectures.losses import FocalLoss  # noqa: F401
