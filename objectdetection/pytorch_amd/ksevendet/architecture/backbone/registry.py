"""Drop-in for ksevendet/architecture/backbone/registry.py:8-93: the model-name registry `KSevenDet._build_backbone`
(ksevendet.py:168-183) resolves `cfg['backbone']` / `cfg['variant']` through."""
import sys
from collections import defaultdict

_module_to_models = defaultdict(set)
_model_to_module = {}
_model_entrypoints = {}


def register_model(fn):
    mod = sys.modules[fn.__module__]
    module_name = fn.__module__.split('.')[-1]
    model_name = fn.__name__
    if hasattr(mod, '__all__'):
        mod.__all__.append(model_name)
    else:
        mod.__all__ = [model_name]
    _model_entrypoints[model_name] = fn
    _model_to_module[model_name] = module_name
    _module_to_models[module_name].add(model_name)
    return fn


def list_models(filter='', module=''):
    import fnmatch
    models = list(_module_to_models[module]) if module else list(_model_entrypoints.keys())
    if filter:
        models = fnmatch.filter(models, filter)
    return sorted(models)


def is_model(model_name):
    return model_name in _model_entrypoints


def model_entrypoint(model_name):
    return _model_entrypoints[model_name]


def list_modules():
    return sorted(_module_to_models.keys())


def is_model_in_modules(model_name, module_names):
    return any(model_name in _module_to_models[n] for n in module_names)
