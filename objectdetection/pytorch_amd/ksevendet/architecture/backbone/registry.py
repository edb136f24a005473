"""Drop-in for ksevendet/architecture/backbone/registry.py:8-93: the model-name registry `KSevenDet._build_backbone`
(ksevendet.py:168-183) resolves `cfg['backbone']` / `cfg['variant']` through.  One `_Registry` object keeps the
three views (entry points, model -> defining module, module -> models); the module-level names the reference exposes
are bound to it."""
import fnmatch
import sys


class _Registry(object):
    def __init__(self):
        self.entrypoints = {}          # model name -> constructor
        self.home = {}                 # model name -> last component of the defining module
        self.by_module = _ModuleIndex()

    def add(self, ctor):
        name, home = ctor.__name__, ctor.__module__.rsplit('.', 1)[-1]
        exported = sys.modules[ctor.__module__].__dict__.setdefault('__all__', [])
        if name not in exported:
            exported.append(name)
        self.entrypoints[name], self.home[name] = ctor, home
        self.by_module[home].add(name)
        return ctor

    def names(self, pattern='', module=''):
        pool = self.by_module[module] if module else self.entrypoints
        return sorted(fnmatch.filter(pool, pattern) if pattern else pool)


class _ModuleIndex(dict):
    """module name -> set of model names; a missing module reads as an empty set (as the reference's defaultdict)."""

    def __missing__(self, key):
        self[key] = set()
        return self[key]


_REG = _Registry()
_model_entrypoints, _model_to_module, _module_to_models = _REG.entrypoints, _REG.home, _REG.by_module
register_model = _REG.add


def list_models(filter='', module=''):
    return _REG.names(filter, module)


def is_model(model_name):
    return model_name in _REG.entrypoints


def model_entrypoint(model_name):
    return _REG.entrypoints[model_name]


def list_modules():
    return sorted(_REG.by_module)


def is_model_in_modules(model_name, module_names):
    return any(model_name in _REG.by_module[m] for m in module_names)
