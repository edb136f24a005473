"""Drop-in for architectures/utils.py (`BBoxTransform` :6-53, `ClipBoxes` :56-71)."""
import torch.nn as nn

from .. import ops_detect


class BBoxTransform(nn.Module):
    def __init__(self, mean=None, std=None):
        super(BBoxTransform, self).__init__()
        self.mean = [0.0, 0.0, 0.0, 0.0] if mean is None else mean
        self.std = [0.1, 0.1, 0.2, 0.2] if std is None else std

    def forward(self, boxes, deltas):
        mean = self.mean.tolist() if hasattr(self.mean, 'tolist') else self.mean
        std = self.std.tolist() if hasattr(self.std, 'tolist') else self.std
        return ops_detect.bbox_decode(boxes, deltas, mean, std)


class ClipBoxes(nn.Module):
    def __init__(self, width=None, height=None):
        super(ClipBoxes, self).__init__()

    def forward(self, boxes, img):
        """Mutates and returns `boxes` (utils.py:61-71)."""
        _, _, height, width = img.shape
        return ops_detect.clip_boxes_(boxes, height, width)
