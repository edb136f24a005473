"""objectdetection.pytorch_amd — MI355X (gfx950) native one-stage-detection hot path.

Drop-in for the reference's `architectures.{retinanet,anchors,losses,utils}` API
(SURVEY.md §8b).  Every numeric op runs in hand-written HIP behind the C ABI declared in
`include/mi355det.h`; there is NO CPU fallback: using an op without the built library or with
a non-GPU tensor raises.
"""
import os as _os

# The launch plans run weight gradients beside data gradients on a second HIP stream.  HIP maps streams
# round-robin onto GPU_MAX_HW_QUEUES (default 4) hardware queues; with RCCL's own streams in the process
# the two compute streams end up sharing one queue and serialise (+17 % step time, measured with
# tools/nccl_rehearse.py).  Read when the HIP runtime starts: effective if set before the first GPU call.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

from . import _lib  # noqa: F401,E402

__all__ = ['_lib']
