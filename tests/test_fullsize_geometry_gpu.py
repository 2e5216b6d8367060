"""Bit-exact geometry parity at the REAL sizes of BASELINE configs[2..4].

The kernel-level files check the voxelizer and the rulebooks on the KITTI grid and on cropped ranges; the key
spaces that differ most from KITTI -- 1024 x 1024 x 40 (nuScenes CBGS, batch 16), 2016 x 2016 x 40 (Lyft CBGS,
batch 6: 1.0e9 cell keys), 512 x 512 x 1 pillars with T = 20 -- are compared with the oracle here, at the
configured samples_per_gpu and max_voxel_num (examples/cbgs/configs/nusc_all_vfev3_spmiddleresnetfhd_rpn2_mghead_syncbn.py,
lyft_all_...py:233-238, examples/point_pillars/configs/nusc_all_point_pillars_mghead_syncbn.py):

  * d3d_voxelize == oracle.clib.points_to_voxel (voxel_oracle.c == the reference's point_cloud_ops.h compiled into
    oracle/_ref, tests/test_oracle_voxel_cpu.py) frame by frame: coordinates, counts, point order, values, fused mean;
  * every rulebook SpMiddleResNetFHD.plan() builds == oracle/spconv_ref.get_indice_pairs, compared as the dense
    neighbour TABLES (strided outputs are in canonical ascending-cell order on both sides, submanifold outputs in input
    order, so nbr[k][o] is directly comparable -- no sorting of 10^7 triples), the input-stationary twin tables and the
    parity-class row plans of the strided data gradients included;
  * the u32 cell-key guard: a batch whose batch * D * H * W does not fit 32 bits is refused, not wrapped.
"""
import os

import numpy as np
import pytest
import torch

from oracle import spconv_ref as R

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# name: in-repo config (field-for-field the reference's), samples_per_gpu as configured, points per frame of the bench
CASES = {
    "nusc_cbgs": dict(config="nusc_cbgs.py", batch=16, points=250000, beams=32, sweeps=10),
    "lyft_cbgs": dict(config="lyft_cbgs.py", batch=6, points=200000, beams=64, sweeps=1),
    "nusc_pp": dict(config="nusc_point_pillars.py", batch=4, points=250000, beams=32, sweeps=10),
}


def _frames(name):
    from det3d_amd import Config, synthetic

    case = CASES[name]
    cfg = Config.fromfile(os.path.join(ROOT, "det3d_amd", "configs", case["config"]))
    ds = synthetic.TaskGroupSynthetic(cfg, case["points"], n_seeds=case["batch"], beams=case["beams"],
                                      sweeps=case["sweeps"])
    frames = [ds.frames[i]["points"] for i in range(case["batch"])]
    return cfg, ds, frames


def _voxelize(ds, frames, mean_channels):
    from det3d_amd.point_cloud_ops import voxelize_batch

    v = ds.vox
    off = np.cumsum([0] + [len(f) for f in frames]).astype(np.int32)
    pts = torch.from_numpy(np.concatenate(frames)).cuda()
    vb = voxelize_batch(pts, torch.from_numpy(off).cuda(), v["voxel_size"], v["pc_range"], v["max_points"],
                        v["max_voxels"], want_voxels=True, mean_channels=mean_channels)
    return vb.sync()


@pytest.mark.parametrize("name", sorted(CASES))
def test_voxelizer_bit_exact_at_configured_size(oracle_lib, name):
    cfg, ds, frames = _frames(name)
    v = ds.vox
    grid = ds.grid  # x, y, z
    expect = {"nusc_cbgs": (1024, 1024, 40, 10, 60000), "lyft_cbgs": (2016, 2016, 40, 10, 80000),
              "nusc_pp": (512, 512, 1, 20, 30000)}[name]
    assert (int(grid[0]), int(grid[1]), int(grid[2]), v["max_points"], v["max_voxels"]) == expect
    c = frames[0].shape[1]
    out = _voxelize(ds, frames, mean_channels=c)
    base, capped = 0, 0
    for b, f in enumerate(frames):
        gv, gc, gn = oracle_lib.points_to_voxel(f, v["voxel_size"], v["pc_range"], v["max_points"], True,
                                                v["max_voxels"])
        m = len(gc)
        capped += m == v["max_voxels"]
        assert int(out["num_voxels"][b]) == m, (name, b)
        co = out["coordinates"][base:base + m].cpu().numpy()
        assert np.all(co[:, 0] == b) and np.array_equal(co[:, 1:], gc), (name, b)
        assert np.array_equal(out["num_points"][base:base + m].cpu().numpy(), gn), (name, b)
        assert np.array_equal(out["voxels"][base:base + m].cpu().numpy(), gv), (name, b)
        mean_ref = gv.sum(1) / gn[:, None].astype(np.float32)  # voxel_encoder.py:206-211
        assert np.allclose(out["mean"][base:base + m].cpu().numpy(), mean_ref, rtol=1e-6, atol=1e-6)
        base += m
    assert base == out["coordinates"].shape[0]
    print("%s: %d frames, %d voxels, %d frames at the max_voxels cap, %d x %d x %d grid" %
          (name, len(frames), base, capped, grid[0], grid[1], grid[2]))


def _ref_tables(pairs, n_in, n_out):
    kvol = len(pairs)
    t_out = np.full((kvol, n_out), -1, np.int32)
    t_in = np.full((kvol, n_in), -1, np.int32)
    for k, (i, o) in enumerate(pairs):
        t_out[k, o] = i
        t_in[k, i] = o
    return t_out, t_in


def _check_row_plan(rb, coords, conv):
    """d3d_rowplan.h: [0] classes, [1..] class starts, [304...] the STABLE permutation of the input rows by parity
    class (i + p) mod s per axis"""
    plan = rb.dgrad_plan.cpu().numpy()
    s, p = np.array(conv.stride), np.array(conv.padding)
    ncls = int(np.prod(s))
    n = len(coords)
    par = (coords[:, 1:].astype(np.int64) + p) % s
    cls = (par[:, 0] * s[1] + par[:, 1]) * s[2] + par[:, 2]
    assert plan[0] == ncls
    starts = np.concatenate([[0], np.cumsum(np.bincount(cls, minlength=ncls))])
    assert np.array_equal(plan[1:2 + ncls], starts)
    assert np.array_equal(plan[304:304 + n], np.argsort(cls, kind="stable"))


@pytest.mark.parametrize("name", ["nusc_cbgs", "lyft_cbgs"])
def test_every_rulebook_of_the_resnet_backbone_bit_exact_at_configured_size(oracle_lib, name):
    from det3d_amd.backbones import SpMiddleResNetFHD

    cfg, ds, frames = _frames(name)
    batch = len(frames)
    out = _voxelize(ds, frames, mean_channels=0)
    coors = out["coordinates"]
    net = SpMiddleResNetFHD(num_input_features=frames[0].shape[1])
    grid = [int(g) for g in ds.grid]
    plan = net.plan_geometry(coors, batch, grid)
    torch.cuda.synchronize()
    rbs = plan["indice_dict"]
    idx = coors.cpu().numpy()
    shape = [grid[2] + 1, grid[1], grid[0]]  # scn.py:357-361: sparse_shape = grid[::-1] + [1, 0, 0]
    assert batch * shape[0] * shape[1] * shape[2] < 2 ** 32
    seen, n_pairs, n_tables = set(), 0, 0
    for m in net.middle_conv.conv_layers():
        key = m._rulebook_key()
        rb = rbs[key]
        if key in seen:
            if not m.subm:
                raise AssertionError("strided layers do not share rulebooks")
            continue
        seen.add(key)
        oi, oshape, pairs = R.get_indice_pairs(idx, batch, shape, m.kernel_size, m.stride, m.padding, m.subm)
        assert rb.n_in == len(idx) and rb.n_out == len(oi), (name, key, rb.n_in, rb.n_out, len(oi))
        assert list(rb.out_shape) == list(oshape)
        t_out, t_in = _ref_tables(pairs, len(idx), len(oi))
        got = rb.nbr_out.view(rb.kvol, rb.cap_out)[:, :rb.n_out].cpu().numpy()
        assert np.array_equal(got, t_out), (name, key, "output-stationary table")
        n_pairs += int((t_out >= 0).sum())
        n_tables += 1
        if not m.subm:
            assert np.array_equal(rb.out_indices.cpu().numpy(), oi), (name, key, "output coordinates")
            got_in = rb.nbr_in.view(rb.kvol, rb.cap_in)[:, :rb.n_in].cpu().numpy()
            assert np.array_equal(got_in, t_in), (name, key, "input-stationary table")
            if rb.dgrad_plan is not None:
                _check_row_plan(rb, idx, m)
            idx, shape = oi, list(oshape)
    assert n_tables == 8  # 4 submanifold levels + 4 strided layers (scn.py:323-355)
    print("%s: %d rulebooks, %d pairs, %d -> %d sites" % (name, n_tables, n_pairs, coors.shape[0], len(idx)))


def test_pillar_canvas_at_configured_size(oracle_lib):
    """nuScenes PointPillars: 512 x 512 x 1 grid, T = 20, M = 30000, batch 4 -- the scatter of per-pillar rows into
    the channels-last canvas (pillar_encoder.py:17-211 -> PointPillarsScatter) against numpy indexing"""
    from det3d_amd.backbones import PointPillarsScatter

    cfg, ds, frames = _frames("nusc_pp")
    out = _voxelize(ds, frames, mean_channels=0)
    co = out["coordinates"]
    feats = torch.randn((co.shape[0], 64), device="cuda")
    sc = PointPillarsScatter(num_input_features=64)
    canvas = sc(feats, co, len(frames), [int(g) for g in ds.grid])
    assert canvas.shape == (4, 64, 512, 512)
    ref = np.zeros((4, 64, 512, 512), np.float32)
    c = co.cpu().numpy()
    ref[c[:, 0], :, c[:, 2], c[:, 3]] = feats.cpu().numpy()
    assert np.array_equal(canvas.cpu().numpy(), ref)


def test_cell_keys_beyond_32_bits_are_refused():
    """rb_lin / the voxelizer's hash keys are uint32: batch * D * H * W must fit (Lyft grid: batch 6 = 1.0e9 fits,
    batch 32 = 5.3e9 does not) -- an explicit capacity error, never a wrapped key"""
    from det3d_amd import _lib, spconv
    from det3d_amd.point_cloud_ops import voxelize_batch

    shape = [41, 2016, 2016]
    idx = torch.tensor([[0, 1, 2, 3], [31, 40, 2015, 2015]], dtype=torch.int32, device="cuda")
    n_dev = torch.tensor([2], dtype=torch.int32, device="cuda")
    ok = spconv.build_subm_rulebook(idx[:1], n_dev - 1, 1, 1, 16, shape, [3, 3, 3])  # 2.7e9 cells: fine
    assert ok.n_out == 1
    with pytest.raises(_lib.D3dError, match="exceed u32 keys"):
        spconv.build_subm_rulebook(idx, n_dev, 2, 2, 32, shape, [3, 3, 3])
    with pytest.raises(_lib.D3dError, match="exceed u32 keys"):
        spconv.build_conv_rulebook(idx, n_dev, 2, 2, 256, shape, [3, 3, 3], [2, 2, 2], [1, 1, 1])
    pts = torch.zeros((4, 4), device="cuda")
    off = torch.arange(0, 33, dtype=torch.int32, device="cuda").clamp_(max=4)
    with pytest.raises(_lib.D3dError, match="exceeds u32 keys"):
        voxelize_batch(pts, off, [0.1, 0.1, 0.15], [-100.8, -100.8, -4.0, 100.8, 100.8, 2.0], 10, 100)
