"""tilings of the f32 (split arithmetic) 64->64 forward kernel (d3d_set_tuning knob 0, variants 50..55) and the f32
weight-gradient kernel on a real KITTI-scale rulebook, one process, HIP events.   python tools/tune_fwd_f32.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from det3d_amd import _lib, spconv, synthetic  # noqa: E402
from det3d_amd.point_cloud_ops import voxelize_batch  # noqa: E402

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
st = _lib.stream_ptr(dev)
f = torch.randn(n, 64, device=dev)
w32 = torch.randn(27, 64, 64, device=dev) * 0.05
wpack = torch.empty((27, 64, 64), device=dev)
_lib.check(L.d3d_pack_conv_weights(w32.data_ptr(), 27, 64, 64, _lib.D3D_F32, 1, wpack.data_ptr(), None, st), "pack")
o = torch.empty(n, 64, device=dev)
balg = P * 64 * 4 + P * 8 + n * 64 * 4 + 27 * 64 * 64 * 4


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


def run(v):
    L.d3d_set_tuning(0, v)
    return L.d3d_sparse_conv_fwd_ex(f.data_ptr(), wpack.data_ptr(), None, rb.nbr_out.data_ptr(), rb.cap_out, n, 27, 64, 64,
                                    _lib.D3D_F32, o.data_ptr(), None, st)


print("forward 64->64 f32 (split), n=%d sites, %d pairs, B_alg %.1f MB" % (n, P, balg / 1e6), flush=True)
names = {0: "default MT2 x8w D1", 50: "MT1 x16w D2", 51: "MT1 x16w D3", 52: "MT2 x8w D2", 53: "MT1 x8w D3", 54: "MT2 x16w D1",
         55: "MT1 x16w D1"}
run(0)
ref = o.clone()
for v in (0, 50, 51, 52, 53, 54, 55):
    o.zero_()
    rc = run(v)
    same = torch.equal(o, ref)
    t = timed(lambda: run(v))
    print("  variant %2d %-20s rc %d  %6.1f us  %6.1f GB/s alg  identical %s" % (v, names[v], rc, t, balg / t / 1e3, same), flush=True)
L.d3d_set_tuning(0, 0)
g = torch.randn(n, 64, device=dev)
dw = torch.empty((27, 64, 64), device=dev)
wsb = L.d3d_sparse_conv_wgrad_workspace_bytes(n, 27, 64, 64, _lib.D3D_F32)
ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
tw = timed(lambda: L.d3d_sparse_conv_wgrad(f.data_ptr(), g.data_ptr(), rb.nbr_out.data_ptr(), rb.cap_out, n, 27, 64, 64,
                                           _lib.D3D_F32, dw.data_ptr(), ws.data_ptr(), wsb, st))
print("wgrad 64x64 f32 (split): %.1f us" % tw, flush=True)
