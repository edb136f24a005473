from .fpn import PANetFPN  # noqa: F401
