"""Type-keyed registries with the contract of det3d/utils/registry.py:6-78 and
det3d/models/registry.py:3-10: classes register themselves under their __name__
(`@REGISTRY.register_module`), and `build_from_cfg(cfg, registry, default_args)` takes a dict
whose "type" names the class (or is the class) while the remaining keys become constructor kwargs."""
import inspect


class Registry:
    """name -> class table."""

    def __init__(self, name):
        self._name = name
        self._classes = {}

    name = property(lambda self: self._name)
    module_dict = property(lambda self: self._classes)

    def __repr__(self):
        return "Registry(name=%s, items=%s)" % (self._name, sorted(self._classes))

    def __contains__(self, key):
        return key in self._classes

    def get(self, key):
        return self._classes.get(key)

    def register_module(self, cls):
        if not inspect.isclass(cls):
            raise TypeError("only classes can be registered, got %s" % type(cls))
        key = cls.__name__
        if key in self._classes:
            raise KeyError("%s is already registered in %s" % (key, self._name))
        self._classes[key] = cls
        return cls


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict) or "type" not in cfg:
        raise TypeError("cfg must be a dict with a 'type' key")
    if default_args is not None and not isinstance(default_args, dict):
        raise TypeError("default_args must be a dict or None")
    kwargs = {k: v for k, v in cfg.items() if k != "type"}
    target = cfg["type"]
    if inspect.isclass(target):
        cls = target
    elif isinstance(target, str):
        cls = registry.get(target)
        if cls is None:
            raise KeyError("%s is not in the %s registry" % (target, registry.name))
    else:
        raise TypeError("type must be a str or a class, got %s" % type(target))
    for key, value in (default_args or {}).items():
        kwargs.setdefault(key, value)
    return cls(**kwargs)


READERS, BACKBONES, NECKS = Registry("reader"), Registry("backbone"), Registry("neck")
HEADS, LOSSES, DETECTORS = Registry("head"), Registry("loss"), Registry("detector")
PIPELINES = Registry("pipeline")  # det3d/datasets/registry.py:3-4
