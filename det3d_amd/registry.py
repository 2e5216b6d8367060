"""Type-keyed registries, same contract as det3d/utils/registry.py:6-78 and
det3d/models/registry.py:3-10: `register_module` decorator, `build_from_cfg(cfg,
registry, default_args)` pops "type" and passes the remaining keys as kwargs."""
import inspect


class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    def __repr__(self):
        return "%s(name=%s, items=%s)" % (type(self).__name__, self._name, list(self._module_dict))

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def register_module(self, cls):
        if not inspect.isclass(cls):
            raise TypeError("module must be a class, but got %s" % type(cls))
        if cls.__name__ in self._module_dict:
            raise KeyError("%s is already registered in %s" % (cls.__name__, self._name))
        self._module_dict[cls.__name__] = cls
        return cls


def build_from_cfg(cfg, registry, default_args=None):
    if not (isinstance(cfg, dict) and "type" in cfg):
        raise TypeError("cfg must be a dict with a 'type' key")
    if not (default_args is None or isinstance(default_args, dict)):
        raise TypeError("default_args must be a dict or None")
    args = dict(cfg)
    kind = args.pop("type")
    if isinstance(kind, str):
        cls = registry.get(kind)
        if cls is None:
            raise KeyError("%s is not in the %s registry" % (kind, registry.name))
    elif inspect.isclass(kind):
        cls = kind
    else:
        raise TypeError("type must be a str or class, got %s" % type(kind))
    for k, v in (default_args or {}).items():
        args.setdefault(k, v)
    return cls(**args)


READERS = Registry("reader")
BACKBONES = Registry("backbone")
NECKS = Registry("neck")
HEADS = Registry("head")
LOSSES = Registry("loss")
DETECTORS = Registry("detector")
