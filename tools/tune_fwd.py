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
names.update({20: "chunk MT2 x8w D3", 21: "chunk MT1 x16w D3", 22: "chunk MT2 x4w D3", 23: "chunk MT2 x8w D1",
              24: "chunk MT4 x4w D3", 25: "chunk MT1 x8w D3", 26: "chunk8 MT2 x8w D2", 27: "chunk6 MT1 x8w D1 3blk", 28: "chunk3 MT1 x4w D1 6blk", 29: "chunk MT1 x8w D1", 30: "group MT2 x8w D3", 31: "group MT1 x16w D3",
              32: "group MT2 x8w D2", 33: "group MT1 x8w D3", 34: "group MT2 x4w D3", 35: "group MT1 x8w D5"})
variants = [int(v) for v in os.environ.get("D3D_TUNE_VARIANTS", "0,23,25,27,28,29,32,33").split(",")]


def run(v, ep=None, rows=None):
    L.d3d_set_tuning(0, v)
    return L.d3d_sparse_conv_fwd_ex(f.data_ptr(), w.data_ptr(), None, rb.nbr_out.data_ptr(), rb.cap_out, rows or n, 27,
                                    64, 64, 1, o.data_ptr(), ctypes.byref(ep) if ep else None, st)


L.d3d_set_tuning(2, 0)
run(0)
ref = o.float().clone()
stats.zero_()
run(0, _lib.ConvEpilogue(stats=stats.data_ptr(), copies=8))
ref_stats = stats.view(8, 128).sum(0).clone()
for v in variants:
    o.zero_()
    _lib.check(run(v), "variant %d" % v)
    err = float((o.float() - ref).abs().max())
    stats.zero_()
    run(v, _lib.ConvEpilogue(stats=stats.data_ptr(), copies=8))
    serr = float(((stats.view(8, 128).sum(0) - ref_stats).abs() / (ref_stats.abs() + 1.0)).max())
    res = []
    for epi in (False, True):
        ep = _lib.ConvEpilogue(stats=stats.data_ptr(), copies=8) if epi else None
        res.append(timed(lambda: run(v, ep)))
    say("  variant %2d %-18s: plain %6.1f us (%.1f %% of 8 TB/s algorithmic), with BN-stats epilogue %6.1f us; "
        "max |out - variant 0| %.3g, stats rel err %.2g" % (v, names[v], res[0], balg / res[0] / 1e3 / 80, res[1], err, serr))
n2 = n // 3
for v in variants:
    us = timed(lambda: run(v, None, n2))
    say("  first %d sites only: variant %2d %-18s: %6.1f us" % (n2, v, names[v], us))
L.d3d_set_tuning(0, -1)
