"""sweep the tilings of spconv_fwd_mfma_kernel<64,64> (d3d_set_tuning knob 0) and the block count of the tile
weight-gradient kernel (knob 1) on a real KITTI-scale rulebook, one process, HIP events.
    python tools/tune_fwd.py [out.txt]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from det3d_amd import _lib, spconv, synthetic  # noqa: E402
from det3d_amd.point_cloud_ops import voxelize_batch  # noqa: E402

out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None


def say(msg):
    print(msg, flush=True)
    if out:
        out.write(msg + "\n")


dev = torch.device("cuda", 0)
ds = synthetic.KittiCarSynthetic(n_points=20000, n_seeds=4)
b = ds.device_batch(0, 4, dev)
vb = voxelize_batch(b["points"], b["frame_offsets"], synthetic.KITTI_CAR["voxel_size"], synthetic.KITTI_CAR["pc_range"],
                    5, 20000, want_voxels=False, mean_channels=4).sync()
x = spconv.SparseConvTensor(vb["mean"], vb["coordinates"], [41, 1600, 1408], 4)
net = spconv.SparseSequential(spconv.SparseConv3d(4, 16, 3, 2, padding=1, bias=False),
                              spconv.SparseConv3d(16, 64, 3, 2, padding=1, bias=False),
                              spconv.SubMConv3d(64, 64, 3, bias=False, indice_key="s2")).to(dev)
net.plan(x)
rb = x.indice_dict["s2"]
n, P = rb.n_out, rb.pairs()
L = _lib.lib()
f = torch.randn(n, 64, device=dev).bfloat16()
w = torch.randn(27, 64, 64, device=dev).bfloat16()
o = torch.empty(n, 64, device=dev, dtype=torch.bfloat16)
stats = torch.zeros(2 * 8 * 64, device=dev)
st = _lib.stream_ptr(dev)
balg = P * 64 * 2 + P * 8 + n * 64 * 2 + 27 * 64 * 64 * 2
names = {0: "default", 1: "MT1 x16w D3", 2: "MT1 x16w D5", 3: "MT2 x8w D3", 4: "MT2 x8w D4", 5: "MT2 x16w D3",
         6: "MT2 x4w D4", 7: "MT4 x4w D2", 8: "MT4 x8w D2", 9: "MT4 x4w D3", 10: "MT1 x8w D4"}


def timed(call, reps=30):
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps


say("forward 64->64 bf16, n=%d sites, %d pairs, B_alg %.1f MB" % (n, P, balg / 1e6))
for remap in (1, 0):
    L.d3d_set_tuning(2, remap)
    for v in (0, 1, 2, 3, 5, 7, 8, 9, 10):
        L.d3d_set_tuning(0, v)
        res = []
        for epi in (False, True):
            ep = _lib.ConvEpilogue(stats=stats.data_ptr(), copies=8) if epi else None
            res.append(timed(lambda: L.d3d_sparse_conv_fwd_ex(f.data_ptr(), w.data_ptr(), None, rb.nbr_out.data_ptr(), rb.cap_out,
                                                              n, 27, 64, 64, 1, o.data_ptr(),
                                                              ctypes.byref(ep) if ep else None, st)))
        say("  xcd_remap %d variant %2d %-12s: plain %6.1f us (%.1f %% of 8 TB/s algorithmic), with BN-stats epilogue %6.1f us" % (
            remap, v, names[v], res[0], balg / res[0] / 1e3 / 80, res[1]))
L.d3d_set_tuning(0, -1)
# a smaller level (the 64->64 layers of the deeper levels see 50-80 k sites): subsample the rulebook's rows
n2 = n // 3
for remap in (1, 0):
    L.d3d_set_tuning(2, remap)
    for v in (0, 1, 3, 5, 8, 10):
        L.d3d_set_tuning(0, v)
        us = timed(lambda: L.d3d_sparse_conv_fwd_ex(f.data_ptr(), w.data_ptr(), None, rb.nbr_out.data_ptr(), rb.cap_out, n2, 27,
                                                    64, 64, 1, o.data_ptr(), None, st))
        say("  first %d sites only: xcd_remap %d variant %2d %-12s: %6.1f us" % (n2, remap, v, names[v], us))
L.d3d_set_tuning(0, -1)
L.d3d_set_tuning(2, 0)
say("weight gradient (tile kernel + reduce), same rulebook")
for ci, co in ((64, 64), (32, 32), (16, 16)):
    ff = torch.randn(n, ci, device=dev).bfloat16()
    d = torch.randn(n, co, device=dev).bfloat16()
    dw = torch.empty(27, ci, co, device=dev)
    for blocks, remap in ((256, 1), (256, 0), (512, 1), (192, 1)):
        L.d3d_set_tuning(1, blocks)
        L.d3d_set_tuning(2, remap)
        wsb = L.d3d_sparse_conv_wgrad_workspace_bytes(n, 27, ci, co, 1)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        us = timed(lambda: L.d3d_sparse_conv_wgrad(ff.data_ptr(), d.data_ptr(), rb.nbr_out.data_ptr(), rb.cap_out, n, 27, ci, co,
                                                   1, dw.data_ptr(), ws.data_ptr(), wsb, st), 20)
        say("  %3dx%-3d blocks %4d xcd_remap %d: %6.1f us (both launches)" % (ci, co, blocks, remap, us))
L.d3d_set_tuning(1, -1)
L.d3d_set_tuning(2, 0)
