"""TEMPORARY: per-wave cycle breakdown of spconv_pairs_kernel through its debug buffer."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import numpy as np
from det3d_amd import _lib, spconv, synthetic
from det3d_amd.point_cloud_ops import voxelize_batch
dev = torch.device("cuda", 0)
ds = synthetic.KittiCarSynthetic(n_points=20000, n_seeds=4)
b = ds.device_batch(0, 4, dev)
vb = voxelize_batch(b["points"], b["frame_offsets"], synthetic.KITTI_CAR["voxel_size"], synthetic.KITTI_CAR["pc_range"], 5, 20000, want_voxels=False, mean_channels=4).sync()
x = spconv.SparseConvTensor(vb["mean"], vb["coordinates"], [41, 1600, 1408], 4)
net = spconv.SparseSequential(
    spconv.SubMConv3d(4, 16, 3, bias=False, indice_key="subm0"), spconv.SparseConv3d(16, 32, 3, 2, padding=1, bias=False),
    spconv.SubMConv3d(32, 32, 3, bias=False, indice_key="subm1"), spconv.SparseConv3d(32, 64, 3, 2, padding=1, bias=False),
    spconv.SubMConv3d(64, 64, 3, bias=False, indice_key="subm2"), spconv.SparseConv3d(64, 64, 3, 2, padding=[0, 1, 1], bias=False),
    spconv.SubMConv3d(64, 64, 3, bias=False, indice_key="subm3")).to(dev)
net.plan(x)
L = _lib.lib(); st = _lib.stream_ptr(dev)
L.d3d_pairs_debug_buffer.argtypes = [ctypes.c_void_p]
for key in ("subm2", "subm3"):
    rb = x.indice_dict[key]; n = rb.n_out; cin = cout = 64
    f = torch.randn(n, cin, device=dev); w32 = torch.randn(27, cin, cout, device=dev) * 0.05
    wpack = torch.empty((27, cin, cout), device=dev)
    _lib.check(L.d3d_pack_conv_weights(w32.data_ptr(), 27, cin, cout, _lib.D3D_F32, 1, wpack.data_ptr(), None, st), "pack")
    o = torch.empty(n, cout, device=dev); table = rb.pair_table("out")
    nslab = (n + 63) // 64 + 16
    dbg = torch.zeros((nslab, 8), dtype=torch.int64, device=dev)
    run = lambda: L.d3d_sparse_conv_fwd_pairs(f.data_ptr(), wpack.data_ptr(), None, table.data_ptr(), n, 27, cin, cout, _lib.D3D_F32, o.data_ptr(), None, st)
    for _ in range(3): run()
    torch.cuda.synchronize()
    L.d3d_pairs_debug_buffer(dbg.data_ptr()); run(); torch.cuda.synchronize(); L.d3d_pairs_debug_buffer(None)
    L.d3d_pairs_ablate.argtypes = [ctypes.c_int]
    def timed(reps=30):
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): run()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1000 / reps
    for fl, what in ((0, "complete"), (1, "no MFMA / split"), (16, "no split"), (2, "no LDS accumulate"), (3, "no MFMA, no LDS acc"), (4, "no gathers (OOB)"), (8, "no barriers"),
                     (7, "no MFMA, LDS acc, gathers"), (15, "nothing but the loop, entries, W publish"), (12, "no gathers no barriers"), (9, "no mfma no barriers")):
        L.d3d_pairs_ablate(fl)
        print("   ablation %2d %-42s %7.1f us" % (fl, what, timed()))
    L.d3d_pairs_ablate(0)
    d = dbg.cpu().numpy()[: (n + 63) // 64]
    tot, bar, ph, T, t0 = d[:, 0], d[:, 1], d[:, 2], d[:, 3], d[:, 4]
    span = (t0 + tot).max() - t0.min()
    print("%s: %d slabs; clock64 ticks (100 MHz?) span %d; per wave: total mean %.0f max %.0f | barrier wait mean %.0f (%.0f%%) | publish+W-load issue mean %.0f (%.0f%%) | tiles mean %.1f max %d; start spread %d" % (
        key, len(d), span, tot.mean(), tot.max(), bar.mean(), 100 * bar.sum() / tot.sum(), ph.mean(), 100 * ph.sum() / tot.sum(), T.mean(), T.max(), t0.max() - t0.min()))
    per_tile = (tot - bar - ph) / T
    print("   non-barrier ticks per tile: mean %.2f p10 %.2f p90 %.2f; corr(total, tiles) %.2f" % (per_tile.mean(), np.percentile(per_tile, 10), np.percentile(per_tile, 90), np.corrcoef(tot, T)[0, 1]))
