"""Import single modules of the Python reference (/root/reference/det3d) in THIS
container, where its third-party deps (numba, spconv, torchvision, addict, boost
extension, ...) are absent.  Used only by make_golden.py to produce golden vectors.

 * every `det3d.*` package is created WITHOUT running its __init__.py (those pull in
   the whole tree); `from det3d.pkg import name` is resolved lazily by scanning the
   package's own .py files for a module that defines `name`;
 * absent third-party modules are replaced by permissive stubs; `numba.jit` /
   `numba.cuda.jit` become identity decorators, so the reference's "kernels" run as
   plain Python over numpy arrays -- slow, but it IS the reference code.
"""
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import os
import sys
import types

import numpy as np

REF_ROOT = "/root/reference"


class _Dummy:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Dummy()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Dummy()


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        full = self.__name__ + "." + name
        if full in sys.modules:
            return sys.modules[full]
        return _Dummy


def _identity_jit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def deco(fn):
        return fn

    return deco


class _LocalArray:
    @staticmethod
    def array(shape, dtype=None):
        return np.zeros(shape, dtype=np.float32 if dtype is None else dtype)


def _install_numba():
    nb = _StubModule("numba")
    nb.jit = _identity_jit
    nb.njit = _identity_jit
    nb.float32 = np.float32
    nb.int32 = np.int32
    nb.prange = range
    cuda = _StubModule("numba.cuda")
    cuda.jit = _identity_jit
    cuda.local = _LocalArray
    cuda.shared = _LocalArray
    nb.cuda = cuda
    sys.modules["numba"] = nb
    sys.modules["numba.cuda"] = cuda


class _AddictDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


STUBS = ["spconv", "spconv.utils", "torchvision", "torchvision.models", "torchvision.models.resnet",
         "cv2", "fire", "pycocotools", "pycocotools.mask", "terminaltables", "tensorboardX", "shapely",
         "shapely.geometry", "nuscenes", "IPython", "skimage", "PIL", "matplotlib",
         "matplotlib.pyplot", "open3d", "vtk", "pyquaternion", "lyft_dataset_sdk"]


COMPILED_EXTS = {"det3d.ops.nms.nms", "det3d.ops.syncbn.syncbn_gpu",
                 "det3d.ops.sigmoid_focal_loss.sigmoid_focal_loss_cuda",
                 "det3d.ops.iou3d.iou3d_cuda", "det3d.ops.pointnet2.PN2",
                 "det3d.ops.point_cloud.point_cloud_ops_cc", "det3d.ops.point_cloud.box_ops_cc"}


class _RefFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if not (fullname == "det3d" or fullname.startswith("det3d.")):
            return None
        rel = fullname.replace(".", "/")
        d = os.path.join(REF_ROOT, rel)
        if os.path.isdir(d):
            spec = importlib.machinery.ModuleSpec(fullname, self, is_package=True)
            spec.submodule_search_locations = [d]
            spec.origin = "refpkg:" + d
            return spec
        f = d + ".py"
        if os.path.isfile(f):
            return importlib.util.spec_from_file_location(fullname, f)
        # compiled extensions of the reference (boost NMS, syncbn, ...) do not exist here
        if fullname in COMPILED_EXTS:
            spec = importlib.machinery.ModuleSpec(fullname, self, is_package=False)
            spec.origin = "refstub"
            return spec
        return None

    def create_module(self, spec):
        if spec.origin == "refstub":
            return _StubModule(spec.name)
        return _LazyPackage(spec.name, spec.submodule_search_locations[0])

    def exec_module(self, module):
        pass


class _LazyPackage(types.ModuleType):
    def __init__(self, name, directory):
        super().__init__(name)
        self.__path__ = [directory]
        self._dir = directory

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        # a submodule or subpackage?
        try:
            return importlib.import_module(self.__name__ + "." + name)
        except Exception:
            pass
        for f in sorted(os.listdir(self._dir)):
            if not f.endswith(".py") or f == "__init__.py":
                continue
            try:
                with open(os.path.join(self._dir, f)) as fh:
                    src = fh.read()
            except OSError:
                continue
            if ("def %s(" % name) not in src and ("class %s(" % name) not in src and \
                    ("class %s:" % name) not in src and ("\n%s = " % name) not in src:
                continue
            try:
                mod = importlib.import_module(self.__name__ + "." + f[:-3])
            except Exception:
                continue
            if hasattr(mod, name):
                return getattr(mod, name)
        # subpackages, one level down
        for d in sorted(os.listdir(self._dir)):
            if os.path.isdir(os.path.join(self._dir, d)) and not d.startswith("_"):
                try:
                    sub = importlib.import_module(self.__name__ + "." + d)
                    return getattr(sub, name)
                except Exception:
                    continue
        raise AttributeError("reference package %s has no %s" % (self.__name__, name))


_INSTALLED = False


def install():
    global _INSTALLED
    if _INSTALLED:
        return
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("%s not present: goldens can only be regenerated in the build "
                           "container" % REF_ROOT)
    _install_numba()
    addict = types.ModuleType("addict")
    addict.Dict = _AddictDict
    sys.modules["addict"] = addict
    for name in STUBS:
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                sys.modules[name] = _StubModule(name)
    sys.meta_path.insert(0, _RefFinder())
    _INSTALLED = True


def ref(modname):
    """import a reference module, e.g. ref('det3d.ops.point_cloud.point_cloud_ops')"""
    install()
    return importlib.import_module(modname)
