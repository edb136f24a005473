"""objectdetection.pytorch_amd — MI355X (gfx950) native one-stage-detection hot path.

Drop-in for the reference's `architectures.{retinanet,anchors,losses,utils}` API
(SURVEY.md §8b).  Every numeric op runs in hand-written HIP behind the C ABI declared in
`include/mi355det.h`; there is NO CPU fallback: using an op without the built library or with
a non-GPU tensor raises.
"""
from . import _lib  # noqa: F401

__all__ = ['_lib']
