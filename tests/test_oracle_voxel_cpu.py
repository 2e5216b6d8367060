"""The CPU oracle (oracle/voxel_oracle.c) pinned against
  (a) goldens produced by running the reference's Python voxelizer itself
      (tests/golden/voxel_cases.npz <- point_cloud_ops.py:7-55,112-184), and
  (b) the reference's compiled C++ twin (oracle/_ref, point_cloud_ops.h:12-71)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

CASES = ["sparse", "capped", "clustered", "pillars"]


def _load(name):
    z = np.load(os.path.join(GOLDEN, "voxel_cases.npz"))
    cfg = z[name + "_cfg"]
    return (z[name + "_points"], cfg[:3].tolist(), cfg[3:9].tolist(), int(cfg[9]), int(cfg[10]),
            z[name + "_voxels"], z[name + "_coors"], z[name + "_num"])


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_python_golden(oracle_lib, name):
    pts, vs, rg, T, M, gv, gc, gn = _load(name)
    v, c, n = oracle_lib.points_to_voxel(pts, vs, rg, T, True, M)
    assert v.shape == gv.shape
    assert np.array_equal(c, gc)
    assert np.array_equal(n, gn)
    assert np.array_equal(v, gv)  # bit exact: voxels are copies of input rows


def test_golden_exercises_cap_and_overflow():
    _, _, _, T, M, gv, _, gn = _load("capped")
    assert gv.shape[0] == M  # the early `break` path was taken
    _, _, _, T, M, _, _, gn = _load("clustered")
    assert gn.max() == T  # more than T points fell into some voxel


@pytest.mark.parametrize("seed", range(6))
def test_oracle_matches_compiled_reference(oracle_lib, seed):
    if oracle_lib.ref_module() is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 4000))
    pts = rng.uniform([-1, -41, -3.2, 0], [71, 41, 1.2, 1], size=(n, 4)).astype(np.float32)
    if seed % 2:
        pts[: n // 2, :3] = pts[0, :3] + rng.normal(0, 0.04, (n // 2, 3)).astype(np.float32)
    T = int(rng.integers(1, 8))
    M = int(rng.choice([50, 400, 20000]))
    vs, rg = [0.05, 0.05, 0.1], [0, -40, -3, 70.4, 40, 1]
    a = oracle_lib.points_to_voxel(pts, vs, rg, T, True, M)
    b = oracle_lib.ref_points_to_voxel(pts, vs, rg, T, M)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_grid_size_matches_reference_rounding(oracle_lib):
    assert oracle_lib.grid_size([0.05, 0.05, 0.1], [0, -40, -3, 70.4, 40, 1]).tolist() == [1408, 1600, 40]
    assert oracle_lib.grid_size([0.16, 0.16, 4], [0, -39.68, -3, 69.12, 39.68, 1]).tolist() == [432, 496, 1]
    assert oracle_lib.grid_size([0.1, 0.1, 0.15], [-100.8, -100.8, -4, 100.8, 100.8, 2]).tolist() == [2016, 2016, 40]


def test_empty_and_all_out_of_range(oracle_lib):
    vs, rg = [0.05, 0.05, 0.1], [0, -40, -3, 70.4, 40, 1]
    v, c, n = oracle_lib.points_to_voxel(np.zeros((0, 4), np.float32), vs, rg, 5, True, 100)
    assert v.shape == (0, 5, 4) and c.shape == (0, 3) and n.shape == (0,)
    pts = np.full((10, 4), 1e3, np.float32)
    v, c, n = oracle_lib.points_to_voxel(pts, vs, rg, 5, True, 100)
    assert v.shape[0] == 0
