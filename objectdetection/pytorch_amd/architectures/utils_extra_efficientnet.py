"""Parameter containers mirroring architectures/utils_extra_efficientnet.py
(`Conv2dStaticSamePadding` :9-52, `MaxPool2dStaticSamePadding` :55-96).  The TF-"same" padding
arithmetic itself runs inside the HIP kernels (ops_effdet.same_pad / LayerSpec.geometry)."""
import torch.nn as nn

from .utils_resnet import _PlanOnly


class Conv2dStaticSamePadding(_PlanOnly):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, bias=True, groups=1, dilation=1, **kwargs):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, bias=bias, groups=groups)
        self.stride = self.conv.stride
        self.kernel_size = self.conv.kernel_size
        self.dilation = self.conv.dilation
        if isinstance(self.stride, int):
            self.stride = [self.stride] * 2
        elif len(self.stride) == 1:
            self.stride = [self.stride[0]] * 2
        if isinstance(self.kernel_size, int):
            self.kernel_size = [self.kernel_size] * 2
        elif len(self.kernel_size) == 1:
            self.kernel_size = [self.kernel_size[0]] * 2


class MaxPool2dStaticSamePadding(_PlanOnly):
    def __init__(self, *args, **kwargs):
        super().__init__()
        self.pool = nn.MaxPool2d(*args, **kwargs)
        self.stride = self.pool.stride
        self.kernel_size = self.pool.kernel_size
