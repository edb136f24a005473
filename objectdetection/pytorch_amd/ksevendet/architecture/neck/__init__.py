from .fpn import PANetFPN  # noqa: F401
from .bifpn import BiFPN  # noqa: F401
