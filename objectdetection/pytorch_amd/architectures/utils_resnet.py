"""Parameter containers mirroring architectures/utils_resnet.py (`BasicBlock` :12-41, `Bottleneck` :44-80): same
attribute names / state_dict keys.  The arithmetic of their forward lives in the fused HIP launch plan (engine.py);
calling a block directly is not supported.  Both blocks are built from one table of (kernel size, output planes)."""
import torch.nn as nn


def conv3x3(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


class _PlanOnly(nn.Module):
    """A module that only HOLDS parameters: its arithmetic is part of the parent's launch plan."""

    def forward(self, x):
        raise RuntimeError('%s is a parameter container of the MI355X launch plan; run the parent '
                           'RetinaNet module instead (no per-module / CPU fallback)' % type(self).__name__)


class _ResidualBlock(_PlanOnly):
    """convN / bnN pairs from `LAYOUT` = ((kernel, planes multiplier), ...); the stride sits on the 3x3 conv (the first
    one of a BasicBlock, the middle one of a Bottleneck), as in torchvision's v1.5 blocks the reference copies."""
    LAYOUT = ()
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(_ResidualBlock, self).__init__()
        width, strided = inplanes, False
        for i, (k, mult) in enumerate(self.LAYOUT, 1):
            s = stride if (k == 3 and not strided) else 1
            strided = strided or k == 3
            setattr(self, 'conv%d' % i, nn.Conv2d(width, planes * mult, kernel_size=k, stride=s, padding=k // 2, bias=False))
            setattr(self, 'bn%d' % i, nn.BatchNorm2d(planes * mult))
            width = planes * mult
        self.relu = nn.ReLU(inplace=True)
        self.downsample, self.stride = downsample, stride


class BasicBlock(_ResidualBlock):
    LAYOUT = ((3, 1), (3, 1))
    expansion = 1


class Bottleneck(_ResidualBlock):
    LAYOUT = ((1, 1), (3, 1), (1, 4))
    expansion = 4
