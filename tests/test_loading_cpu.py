"""f2: LoadPointCloudFromFile (det3d_amd/loading.py) against the reference's step run on the same synthetic files
(tests/golden/loading.npz): key frame + randomly chosen sweeps with the near-sensor points removed, the 4x4 transform
into the key frame and the time-lag column; KITTI with / without a `_reduced` copy; Lyft."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

from det3d_amd import loading


@pytest.fixture()
def files(tmp_path):
    z = np.load(os.path.join(GOLDEN, "loading.npz"))
    z["nusc_key"].tofile(tmp_path / "key.bin")
    sweeps = []
    for i in range(6):
        z["nusc_sweep%d_raw" % i].tofile(tmp_path / ("sweep%d.bin" % i))
        sweeps.append(dict(lidar_path=str(tmp_path / ("sweep%d.bin" % i)),
                           transform_matrix=z["nusc_sweep%d_matrix" % i] if i != 4 else None,
                           time_lag=float(z["nusc_sweep%d_lag" % i])))
    os.makedirs(tmp_path / "training" / "velodyne")
    z["kitti_raw"].tofile(tmp_path / "training" / "velodyne" / "000001.bin")
    return z, tmp_path, sweeps


@pytest.mark.parametrize("nsweeps", [1, 4])
def test_nuscenes_sweeps_match_the_reference_step(files, nsweeps):
    z, tmp, sweeps = files
    step = loading.LoadPointCloudFromFile(dataset="NuScenesDataset", device="cpu")
    np.random.seed(9)
    res, _ = step({"lidar": {"nsweeps": nsweeps}, "metadata": {}}, dict(lidar_path=str(tmp / "key.bin"), sweeps=sweeps))
    assert res["type"] == "NuScenesDataset"
    for k in ("points", "times", "combined"):
        got, ref = res["lidar"][k].numpy(), z["nusc_n%d_%s" % (nsweeps, k)]
        assert got.shape == ref.shape and got.dtype == np.float32, k
        assert np.allclose(got, ref, atol=1e-5), (k, np.abs(got - ref).max())
    with pytest.raises(AssertionError):
        step({"lidar": {"nsweeps": 8}, "metadata": {}}, dict(lidar_path=str(tmp / "key.bin"), sweeps=sweeps))


def test_kitti_prefers_the_reduced_cloud_and_lyft_reads_the_top_lidar(files):
    z, tmp, _ = files
    step = loading.LoadPointCloudFromFile(dataset="KittiDataset", device="cpu")
    info = dict(point_cloud=dict(velodyne_path="training/velodyne/000001.bin"))
    meta = {"image_prefix": str(tmp), "num_point_features": 4}
    res, _ = step({"lidar": {}, "metadata": dict(meta)}, info)
    assert np.array_equal(res["lidar"]["points"].numpy(), z["kitti_points"])
    os.makedirs(tmp / "training" / "velodyne_reduced")
    z["kitti_raw"][:200].tofile(tmp / "training" / "velodyne_reduced" / "000001.bin")
    res, _ = step({"lidar": {}, "metadata": dict(meta)}, info)
    assert np.array_equal(res["lidar"]["points"].numpy(), z["kitti_reduced_points"])
    lyft = loading.LoadPointCloudFromFile(dataset="LyftDataset", device="cpu")
    res, _ = lyft({"lidar": {}, "metadata": {}}, dict(ref_info=dict(LIDAR_TOP=dict(lidar_path=str(tmp / "key.bin")))))
    assert np.array_equal(res["lidar"]["points"].numpy(), z["lyft_points"])
    with pytest.raises(NotImplementedError):
        loading.LoadPointCloudFromFile(dataset="Other", device="cpu")({"lidar": {}, "metadata": {}}, {})
