"""TEST INFRASTRUCTURE.  Drives every entry point of the sanitizer build of the oracle (libd3d_oracle_asan.so) on
ragged / empty / capped / adversarial inputs.  Run by tests/test_oracle_sanitizers_cpu.py in a child process with
libasan preloaded; any AddressSanitizer / UBSan report aborts the process."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import build as _build  # noqa: E402
from oracle import clib  # noqa: E402


def main():
    _build.LIB = os.path.join(HERE, "libd3d_oracle_asan.so")  # make clib bind the sanitized library
    clib._LIB = None
    L = clib.lib()
    assert "asan" in L._name
    rng = np.random.default_rng(0)
    # voxelizer: empty, all out of range, > T points per voxel, voxel cap (the reference's `break`), 1-cell grid
    vs, rg = [0.5, 0.5, 1.0], [0, -4, -1, 8, 4, 1]
    for n, T, M in ((0, 3, 10), (1, 1, 1), (500, 2, 7), (5000, 5, 100000), (300, 35, 20000)):
        pts = rng.uniform([-1, -5, -2, 0], [9, 5, 2, 1], (n, 4)).astype(np.float32)
        v, c, k = clib.points_to_voxel(pts, vs, rg, T, True, M)
        assert len(v) == len(c) == len(k) <= M and (k <= T).all()
    clib.points_to_voxel(np.full((50, 4), 3.0, np.float32), vs, rg, 4, True, 5)          # every point in one cell
    clib.points_to_voxel(rng.random((20, 4)).astype(np.float32), [8, 8, 2], rg, 3, True, 5)  # 1-cell grid
    # rotated NMS / IoU: empty, single, duplicates, degenerate (zero-size) boxes
    for n in (0, 1, 2, 65, 300):
        d = np.concatenate([rng.uniform(0, 10, (n, 2)), rng.uniform(0.5, 4, (n, 2)), rng.uniform(-4, 4, (n, 1)),
                            rng.random((n, 1))], 1).astype(np.float32)
        if n > 2:
            d[1] = d[0]
            d[2, 2:4] = 0.0
        clib.rotate_nms_cc(d, 0.3)
        clib.rotate_iou(d[:, :5], d[:, :5])
        clib.box_corners(d[:, :5])
        clib.nms_sorted(np.concatenate([d[:, :2], d[:, :2] + d[:, 2:4]], 1), 0.5)
    # iou3d + evaluation IoU: identical boxes (8+ intersection points), touching, nested, empty
    for n in (0, 1, 40):
        c = rng.uniform(0, 6, (n, 2))
        s = rng.uniform(0.5, 3, (n, 2))
        b = np.concatenate([c - s / 2, c + s / 2, rng.uniform(-3.2, 3.2, (n, 1))], 1).astype(np.float32)
        if n > 3:
            b[1] = b[0]
            b[2, 4] = b[0, 4] + np.float32(np.pi / 2)
            b[3] = [0, 0, 0, 0, 0]
        for mode in ("overlap", "iou_bev", "iou_normal"):
            clib.iou3d_pairwise(b, b, mode)
        clib.iou3d_nms(b, 0.1, False)
        clib.iou3d_nms(b, 0.1, True)
        r = np.concatenate([c, s, rng.uniform(-3.2, 3.2, (n, 1))], 1).astype(np.float32)
        if n > 1:
            r[1] = r[0]
        for crit in (-1, 0, 1, 2):
            clib.rotate_iou_eval(r, r, crit)
    print("sanitize_run: ok")


if __name__ == "__main__":
    main()
