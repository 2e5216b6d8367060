"""Only spconv_fwd_mfma_kernel<64,64> (bf16, with the BN-statistics epilogue as in the step) on a real KITTI-scale
submanifold rulebook: the workload the SQ / TCP / TCC counter passes of tools/pmc_fwd.sh profile.
    python tools/pmc_fwd.py [launches]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from det3d_amd import _lib, spconv, synthetic  # noqa: E402
from det3d_amd.point_cloud_ops import voxelize_batch  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
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
ep = _lib.ConvEpilogue(stats=stats.data_ptr(), copies=8)
import ctypes  # noqa: E402

for i in range(3 + reps):
    rc = L.d3d_sparse_conv_fwd_ex(f.data_ptr(), w.data_ptr(), None, rb.nbr_out.data_ptr(), rb.cap_out, n, 27, 64, 64,
                                  _lib.D3D_BF16, o.data_ptr(), ctypes.byref(ep), st)
    _lib.check(rc, "fwd")
torch.cuda.synchronize()
print("n=%d pairs=%d launches=%d B_alg=%.1f MB" % (n, P, 3 + reps, (P * 64 * 2 + P * 8 + n * 64 * 2 + 27 * 64 * 64 * 2) / 1e6))
