"""Property tests (hypothesis) of the voxelizer oracle against the reference's compiled C++ twin
(oracle/_ref, point_cloud_ops.h:12-71): random clouds with out-of-range points, > T points per
voxel, > M voxels (the early `break`), points exactly on cell / range boundaries."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

VS, RG = [0.05, 0.05, 0.1], [0, -40.0, -3.0, 70.4, 40.0, 1.0]


@st.composite
def clouds(draw):
    n = draw(st.integers(0, 600))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    mode = draw(st.sampled_from(["uniform", "clustered", "boundary", "tiny_box"]))
    rng = np.random.default_rng(seed)
    if mode == "uniform":
        p = rng.uniform([-3, -43, -3.5, 0], [73, 43, 1.5, 1], size=(n, 4))
    elif mode == "clustered":
        c = rng.uniform([1, -30, -2.5, 0], [60, 30, 0.5, 1], size=(max(n // 20, 1), 4))
        p = c[rng.integers(0, len(c), n)] + rng.normal(0, 0.04, (n, 4))
    elif mode == "boundary":
        # coordinates that are exact multiples of the voxel size or the range limits
        ix = rng.integers(-2, 1410, n)
        iy = rng.integers(-2, 1602, n)
        iz = rng.integers(-2, 42, n)
        p = np.stack([ix * 0.05, -40 + iy * 0.05, -3 + iz * 0.1, rng.uniform(0, 1, n)], 1)
    else:
        p = rng.uniform([10, 0, -1, 0], [10.3, 0.3, -0.7, 1], size=(n, 4))
    T = draw(st.integers(1, 6))
    M = draw(st.sampled_from([1, 7, 40, 20000]))
    return p.astype(np.float32), T, M


@settings(max_examples=60, deadline=None)
@given(clouds())
def test_oracle_equals_compiled_reference(oracle_lib, case):
    if oracle_lib.ref_module() is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    pts, T, M = case
    a = oracle_lib.points_to_voxel(pts, VS, RG, T, True, M)
    b = oracle_lib.ref_points_to_voxel(pts, VS, RG, T, M)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    v, c, n = a
    assert len(c) <= M and (len(n) == 0 or (n.min() >= 1 and n.max() <= T))
