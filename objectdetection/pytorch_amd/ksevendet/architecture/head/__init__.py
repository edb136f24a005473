from .simple_head import Classifier, Regressor  # noqa: F401
