"""ctypes bindings of oracle/libd3d_oracle.so (plain-C restatements) and the
optional oracle/_ref module (the reference's own C++ compiled here)."""
import ctypes
import importlib.util
import os

import numpy as np

from . import build as _build

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = _build.LIB
        if not os.path.exists(path):
            _build.build_oracle()
        L = ctypes.CDLL(path)
        f32p = ctypes.POINTER(ctypes.c_float)
        i32p = ctypes.POINTER(ctypes.c_int32)
        f64p = ctypes.POINTER(ctypes.c_double)
        L.d3d_oracle_grid_size.argtypes = [f32p, f32p, i32p]
        L.d3d_oracle_grid_size.restype = None
        L.d3d_oracle_points_to_voxel.argtypes = [f32p, ctypes.c_int64, ctypes.c_int, f32p, f32p,
                                                 ctypes.c_int, ctypes.c_int, f32p, i32p, i32p]
        L.d3d_oracle_points_to_voxel.restype = ctypes.c_int
        L.d3d_oracle_box_corners.argtypes = [f32p, ctypes.c_int, f32p]
        L.d3d_oracle_box_corners.restype = None
        L.d3d_oracle_rotate_iou.argtypes = [f32p, ctypes.c_int, f32p, ctypes.c_int, f64p]
        L.d3d_oracle_rotate_iou.restype = None
        L.d3d_oracle_rotate_nms.argtypes = [f32p, i32p, ctypes.c_int, ctypes.c_float, i32p, f64p]
        L.d3d_oracle_rotate_nms.restype = ctypes.c_int
        L.d3d_oracle_nms.argtypes = [f32p, ctypes.c_int, ctypes.c_float, i32p]
        L.d3d_oracle_nms.restype = ctypes.c_int
        L.d3d_oracle_iou3d_pairwise.argtypes = [f32p, ctypes.c_int, f32p, ctypes.c_int, ctypes.c_int, f32p]
        L.d3d_oracle_iou3d_pairwise.restype = None
        L.d3d_oracle_iou3d_nms.argtypes = [f32p, ctypes.c_int, ctypes.c_float, ctypes.c_int, i32p, f32p]
        L.d3d_oracle_iou3d_nms.restype = ctypes.c_int
        L.d3d_oracle_rotate_iou_eval.argtypes = [f32p, ctypes.c_int, f32p, ctypes.c_int, ctypes.c_int, f32p]
        L.d3d_oracle_rotate_iou_eval.restype = None
        _LIB = L
    return _LIB


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def ref_module():
    """The compiled REFERENCE C++ (oracle/_ref), or None if it was never built."""
    path = _build.ref_module_path()
    if not os.path.exists(path):
        path = _build.build_ref()
    if not path or not os.path.exists(path):
        return None
    spec = importlib.util.spec_from_file_location("d3d_ref", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ------------------------------------------------------------------ voxelizer
def grid_size(voxel_size, coors_range):
    vs = np.ascontiguousarray(voxel_size, dtype=np.float32)
    rg = np.ascontiguousarray(coors_range, dtype=np.float32)
    g = np.zeros(3, dtype=np.int32)
    lib().d3d_oracle_grid_size(_p(vs, ctypes.c_float), _p(rg, ctypes.c_float), _p(g, ctypes.c_int32))
    return g


def points_to_voxel(points, voxel_size, coors_range, max_points=35, reverse_index=True,
                    max_voxels=20000):
    """Same signature/returns as det3d/ops/point_cloud/point_cloud_ops.py:112-184."""
    assert reverse_index, "only the zyx (reverse) variant is on the hot path"
    points = np.ascontiguousarray(points, dtype=np.float32)
    vs = np.ascontiguousarray(voxel_size, dtype=np.float32)
    rg = np.ascontiguousarray(coors_range, dtype=np.float32)
    n, c = points.shape
    voxels = np.empty((max_voxels, max_points, c), dtype=np.float32)
    coors = np.empty((max_voxels, 3), dtype=np.int32)
    num = np.empty((max_voxels,), dtype=np.int32)
    m = lib().d3d_oracle_points_to_voxel(_p(points, ctypes.c_float), n, c, _p(vs, ctypes.c_float),
                                         _p(rg, ctypes.c_float), max_points, max_voxels,
                                         _p(voxels, ctypes.c_float), _p(coors, ctypes.c_int32),
                                         _p(num, ctypes.c_int32))
    if m < 0:
        raise MemoryError("oracle voxelizer: dense grid allocation failed")
    return voxels[:m], coors[:m], num[:m]


def ref_points_to_voxel(points, voxel_size, coors_range, max_points=35, max_voxels=20000):
    """The reference's compiled C++ twin (point_cloud_ops.h:12-71) behind the same
    wrapper logic as point_cloud_ops.py:141-184.  Needs oracle/_ref."""
    mod = ref_module()
    if mod is None:
        raise RuntimeError("oracle/_ref is not built (needs /root/reference)")
    points = np.ascontiguousarray(points, dtype=np.float32)
    vs = np.asarray(voxel_size, dtype=np.float32)
    rg = np.asarray(coors_range, dtype=np.float32)
    shape = tuple(np.round((rg[3:] - rg[:3]) / vs).astype(np.int32).tolist())[::-1]
    num = np.zeros((max_voxels,), dtype=np.int32)
    grid = -np.ones(shape, dtype=np.int32)
    voxels = np.zeros((max_voxels, max_points, points.shape[-1]), dtype=np.float32)
    coors = np.zeros((max_voxels, 3), dtype=np.int32)
    m = mod.points_to_voxel_3d_f32(points, voxels, coors, num, grid, vs.tolist(), rg.tolist(),
                                   max_points, max_voxels)
    return voxels[:m], coors[:m], num[:m]


# ------------------------------------------------------------------------ nms
def box_corners(boxes):
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    out = np.empty((boxes.shape[0], 4, 2), dtype=np.float32)
    lib().d3d_oracle_box_corners(_p(boxes, ctypes.c_float), boxes.shape[0], _p(out, ctypes.c_float))
    return out


def rotate_iou(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    out = np.empty((a.shape[0], b.shape[0]), dtype=np.float64)
    lib().d3d_oracle_rotate_iou(_p(a, ctypes.c_float), a.shape[0], _p(b, ctypes.c_float),
                                b.shape[0], _p(out, ctypes.c_double))
    return out


def rotate_nms_cc(dets, thresh, return_margin=False):
    """dets [n,6] = (x, y, w, l, r, score); returns kept indices like
    det3d/ops/nms/nms_cpu.py:34-45."""
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    n = dets.shape[0]
    order = dets[:, 5].argsort()[::-1].astype(np.int32)
    order = np.ascontiguousarray(order)
    boxes = np.ascontiguousarray(dets[:, :5])
    keep = np.empty((max(n, 1),), dtype=np.int32)
    margin = ctypes.c_double(0.0)
    k = lib().d3d_oracle_rotate_nms(_p(boxes, ctypes.c_float), _p(order, ctypes.c_int32), n,
                                    float(thresh), _p(keep, ctypes.c_int32), ctypes.byref(margin))
    res = keep[:k].tolist()
    return (res, margin.value) if return_margin else res


def nms_sorted(boxes, thresh):
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    keep = np.empty((max(boxes.shape[0], 1),), dtype=np.int32)
    k = lib().d3d_oracle_nms(_p(boxes, ctypes.c_float), boxes.shape[0], float(thresh),
                             _p(keep, ctypes.c_int32))
    return keep[:k].tolist()


# ---------------------------------------------------------------------- iou3d
def ref_iou3d_module():
    """The reference's iou3d device functions compiled for the host (oracle/_ref), or None."""
    path = _build.ref_iou3d_module_path()
    if not os.path.exists(path):
        path = _build.build_ref_iou3d()
    if not path or not os.path.exists(path):
        return None
    spec = importlib.util.spec_from_file_location("d3d_ref_iou3d", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def iou3d_pairwise(a, b, mode):
    """mode: "overlap" | "iou_bev" | "iou_normal"; boxes (x1, y1, x2, y2, ry)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    out = np.empty((a.shape[0], b.shape[0]), dtype=np.float32)
    lib().d3d_oracle_iou3d_pairwise(_p(a, ctypes.c_float), a.shape[0], _p(b, ctypes.c_float), b.shape[0],
                                    {"overlap": 0, "iou_bev": 1, "iou_normal": 2}[mode], _p(out, ctypes.c_float))
    return out


def iou3d_nms(boxes_sorted, thresh, normal=False, return_margin=False):
    boxes = np.ascontiguousarray(boxes_sorted, dtype=np.float32)
    n = boxes.shape[0]
    keep = np.empty((max(n, 1),), dtype=np.int32)
    margin = ctypes.c_float(0.0)
    k = lib().d3d_oracle_iou3d_nms(_p(boxes, ctypes.c_float), n, float(thresh), int(normal),
                                   _p(keep, ctypes.c_int32), ctypes.byref(margin))
    res = keep[:k].tolist()
    return (res, margin.value) if return_margin else res


def rotate_iou_eval(boxes, query_boxes, criterion=-1):
    """rotate_iou_gpu_eval (kitti_object_eval_python/rotate_iou.py:305-341): [N, K]."""
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    q = np.ascontiguousarray(query_boxes, dtype=np.float32)
    out = np.zeros((boxes.shape[0], q.shape[0]), dtype=np.float32)
    if out.size:
        lib().d3d_oracle_rotate_iou_eval(_p(boxes, ctypes.c_float), boxes.shape[0], _p(q, ctypes.c_float),
                                         q.shape[0], int(criterion), _p(out, ctypes.c_float))
    return out
