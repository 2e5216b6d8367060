"""Only the f32 (split arithmetic) 64 -> 64 submanifold forward on the benched KITTI geometry (subm2: 126k sites, 13 pairs
per site), `reps` launches under d3d_set_tuning knob 11 = argv[1] (0: spconv_fwd_mfma_kernel, 1..3: the chunk-resident
tilings, 100: the pair-compacted kernel d3d_sparse_conv_fwd_pairs) -- the workload tools/pmc_fwd32.sh collects SQ / TA / TCP counters on.
    python tools/pmc_fwd32.py <knob 11 value> [reps] [key] [cin] [cout]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from det3d_amd import _lib, spconv, synthetic  # noqa: E402
from det3d_amd.point_cloud_ops import voxelize_batch  # noqa: E402

v = int(sys.argv[1]) if len(sys.argv) > 1 else 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
key = sys.argv[3] if len(sys.argv) > 3 else "subm2"
cin = int(sys.argv[4]) if len(sys.argv) > 4 else 64
cout = int(sys.argv[5]) if len(sys.argv) > 5 else 64
dev = torch.device("cuda", 0)
ds = synthetic.KittiCarSynthetic(n_points=20000, n_seeds=4)
b = ds.device_batch(0, 4, dev)
vb = voxelize_batch(b["points"], b["frame_offsets"], synthetic.KITTI_CAR["voxel_size"], synthetic.KITTI_CAR["pc_range"],
                    5, 20000, want_voxels=False, mean_channels=4).sync()
x = spconv.SparseConvTensor(vb["mean"], vb["coordinates"], [41, 1600, 1408], 4)
net = spconv.SparseSequential(
    spconv.SubMConv3d(4, 16, 3, bias=False, indice_key="subm0"),
    spconv.SparseConv3d(16, 32, 3, 2, padding=1, bias=False),
    spconv.SubMConv3d(32, 32, 3, bias=False, indice_key="subm1"),
    spconv.SparseConv3d(32, 64, 3, 2, padding=1, bias=False),
    spconv.SubMConv3d(64, 64, 3, bias=False, indice_key="subm2"),
    spconv.SparseConv3d(64, 64, 3, 2, padding=[0, 1, 1], bias=False),
    spconv.SubMConv3d(64, 64, 3, bias=False, indice_key="subm3")).to(dev)
net.plan(x)
rb = x.indice_dict[key]
n, P = rb.n_out, rb.pairs()
L = _lib.lib()
st = _lib.stream_ptr(dev)
f = torch.randn(n, cin, device=dev)
w32 = torch.randn(27, cin, cout, device=dev) * 0.05
wpack = torch.empty((27, cin, cout), device=dev)
_lib.check(L.d3d_pack_conv_weights(w32.data_ptr(), 27, cin, cout, _lib.D3D_F32, 1, wpack.data_ptr(), None, st), "pack")
o = torch.empty(n, cout, device=dev)
if v == 100:  # the pair-compacted kernel (round 6)
    table = rb.pair_table("out")
    for _ in range(reps):
        _lib.check(L.d3d_sparse_conv_fwd_pairs(f.data_ptr(), wpack.data_ptr(), None, table.data_ptr(), n, 27, cin, cout,
                                               _lib.D3D_F32, o.data_ptr(), None, st), "pairs")
    torch.cuda.synchronize()
    print("pairs kernel %d->%d %s n=%d pairs=%d launches=%d" % (cin, cout, key, n, P, reps))
    sys.exit(0)
L.d3d_set_tuning(11, v)
for _ in range(reps):
    _lib.check(L.d3d_sparse_conv_fwd_ex(f.data_ptr(), wpack.data_ptr(), None, rb.nbr_out.data_ptr(), rb.cap_out, n, 27, cin,
                                        cout, _lib.D3D_F32, o.data_ptr(), None, st), "fwd")
torch.cuda.synchronize()
print("knob11=%d %d->%d %s n=%d pairs=%d launches=%d" % (v, cin, cout, key, n, P, reps))
