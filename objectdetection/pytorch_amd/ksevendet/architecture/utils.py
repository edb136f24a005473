"""ksevendet/architecture/utils.py:26-71 = architectures/utils.py (`BBoxTransform`, `ClipBoxes`)."""
from ...architectures.utils import BBoxTransform, ClipBoxes  # noqa: F401
