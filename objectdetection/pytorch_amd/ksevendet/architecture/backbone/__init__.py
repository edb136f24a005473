from . import registry, resnet  # noqa: F401
