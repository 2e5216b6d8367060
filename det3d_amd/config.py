"""Python-module configs, as det3d/torchie/utils/config.py:51-162 (`Config.fromfile`,
attribute access on nested dicts) without the `addict` dependency."""
import importlib.util
import os
import sys


class ConfigDict(dict):
    """dict with attribute access; nested dicts are converted on the way in."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, ConfigDict):
            return v
        if isinstance(v, dict):
            return ConfigDict(v)
        if isinstance(v, list):
            return [ConfigDict._wrap(x) for x in v]
        if isinstance(v, tuple):
            return tuple(ConfigDict._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, ConfigDict._wrap(v))

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError("'%s' object has no attribute '%s'" % (type(self).__name__, name))

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        del self[name]

    def copy(self):
        return ConfigDict(self)

    def setdefault(self, k, d=None):
        if k not in self:
            self[k] = d
        return self[k]

    def update(self, *a, **k):
        for kk, v in dict(*a, **k).items():
            self[kk] = v

    def to_dict(self):
        def un(v):
            if isinstance(v, dict):
                return {k: un(x) for k, x in v.items()}
            if isinstance(v, (list, tuple)):
                return type(v)(un(x) for x in v)
            return v

        return un(self)


class Config:
    @staticmethod
    def fromfile(filename):
        filename = os.path.abspath(os.path.expanduser(filename))
        if not os.path.isfile(filename):
            raise FileNotFoundError('file "%s" does not exist' % filename)
        if not filename.endswith(".py"):
            raise IOError("Only .py configs are supported")
        name = os.path.basename(filename)[:-3]
        if "." in name:
            raise ValueError("Dots are not allowed in config file path.")
        from . import compat

        compat.install()  # configs do `from det3d.builder import build_box_coder`
        cfg_dir = os.path.dirname(filename)
        sys.path.insert(0, cfg_dir)
        try:
            spec = importlib.util.spec_from_file_location("_d3d_cfg_" + name, filename)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
        finally:
            sys.path.pop(0)
        cfg = {k: v for k, v in vars(mod).items() if not k.startswith("__")}
        return Config(cfg, filename=filename)

    def __init__(self, cfg_dict=None, filename=None):
        cfg_dict = {} if cfg_dict is None else cfg_dict
        if not isinstance(cfg_dict, dict):
            raise TypeError("cfg_dict must be a dict, but got %s" % type(cfg_dict))
        import types

        clean = {k: v for k, v in cfg_dict.items() if not isinstance(v, types.ModuleType)
                 and not isinstance(v, types.FunctionType)}
        object.__setattr__(self, "_cfg_dict", ConfigDict(clean))
        object.__setattr__(self, "_filename", filename)
        text = ""
        if filename:
            with open(filename) as f:
                text = f.read()
        object.__setattr__(self, "_text", text)

    @property
    def filename(self):
        return self._filename

    @property
    def text(self):
        return self._text

    def __repr__(self):
        return "Config (path: %s): %r" % (self._filename, self._cfg_dict)

    def __len__(self):
        return len(self._cfg_dict)

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setattr__(self, name, value):
        self._cfg_dict[name] = value

    def __setitem__(self, name, value):
        self._cfg_dict[name] = value

    def __iter__(self):
        return iter(self._cfg_dict)

    def __contains__(self, k):
        return k in self._cfg_dict
