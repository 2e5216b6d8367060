"""spconv_fwd_resident_kernel (d3d_set_tuning knob 11) against spconv_fwd_mfma_kernel on the submanifold rulebooks of the
benched KITTI batch (4 frames x 20k points through SpMiddleFHD's geometry): f32 features in split arithmetic, every
channel pair of the backbone, plain and with the BatchNorm-statistics epilogue; HIP events, one process.

    python tools/tune_resident.py [out.txt]

Prints per (channel pair, rulebook, tiling set): microseconds, algorithmic GB/s (SURVEY 8(d) bytes), whether the produced
rows are bit-identical to the per-offset kernel's and how far the statistics are from its statistics."""
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
        out.flush()


dev = torch.device("cuda", 0)
ds = synthetic.KittiCarSynthetic(n_points=20000, n_seeds=4)
b = ds.device_batch(0, 4, dev)
vb = voxelize_batch(b["points"], b["frame_offsets"], synthetic.KITTI_CAR["voxel_size"], synthetic.KITTI_CAR["pc_range"],
                    5, 20000, want_voxels=False, mean_channels=4).sync()
x = spconv.SparseConvTensor(vb["mean"], vb["coordinates"], [41, 1600, 1408], 4)
# the geometry of SpMiddleFHD (scn.py:106-157): one submanifold key per level
net = spconv.SparseSequential(
    spconv.SubMConv3d(4, 16, 3, bias=False, indice_key="subm0"),
    spconv.SparseConv3d(16, 32, 3, 2, padding=1, bias=False),
    spconv.SubMConv3d(32, 32, 3, bias=False, indice_key="subm1"),
    spconv.SparseConv3d(32, 64, 3, 2, padding=1, bias=False),
    spconv.SubMConv3d(64, 64, 3, bias=False, indice_key="subm2"),
    spconv.SparseConv3d(64, 64, 3, 2, padding=[0, 1, 1], bias=False),
    spconv.SubMConv3d(64, 64, 3, bias=False, indice_key="subm3")).to(dev)
net.plan(x)
L = _lib.lib()
st = _lib.stream_ptr(dev)


def timed(call, reps=40):
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


# (cin, cout, rulebook key): the submanifold layers of the backbone and, on the same tables, the channel pairs of its
# strided layers (their own tables differ in fill, not in what the kernel does per stage)
cases = [(16, 16, "subm0"), (16, 32, "subm0"), (32, 16, "subm1"), (32, 32, "subm1"), (32, 64, "subm1"), (64, 32, "subm2"),
         (64, 64, "subm2"), (64, 64, "subm3")]
sets = [int(v) for v in os.environ.get("D3D_TUNE_SETS", "0,1,2,3").split(",")]
for cin, cout, key in cases:
    rb = x.indice_dict[key]
    n, P = rb.n_out, rb.pairs()
    f = torch.randn(n, cin, device=dev)
    w32 = torch.randn(27, cin, cout, device=dev) * 0.05
    wpack = torch.empty((27, cin, cout), device=dev)
    _lib.check(L.d3d_pack_conv_weights(w32.data_ptr(), 27, cin, cout, _lib.D3D_F32, 1, wpack.data_ptr(), None, st), "pack")
    o = torch.empty(n, cout, device=dev)
    nws = L.d3d_conv_epilogue_ws_floats(n, cout)
    ws = torch.zeros(nws, device=dev)
    balg = P * cin * 4 + 27 * n * 4 + n * cout * 4
    say("%d -> %d on %s: n = %d sites, %d pairs (%.2f per site), algorithmic %.1f MB" % (cin, cout, key, n, P, P / n, balg / 1e6))

    def run(v, stats=False):
        L.d3d_set_tuning(11, v)
        ep = _lib.ConvEpilogue(stats=ws.data_ptr(), ws_floats=nws) if stats else None
        return L.d3d_sparse_conv_fwd_ex(f.data_ptr(), wpack.data_ptr(), None, rb.nbr_out.data_ptr(), rb.cap_out, n, 27, cin,
                                        cout, _lib.D3D_F32, o.data_ptr(), ctypes.byref(ep) if ep else None, st)

    run(0, True)
    torch.cuda.synchronize()
    ref, ref_stats = o.clone(), ws[:2 * cout].clone()
    for v in sets:
        o.zero_()
        ws.zero_()
        rc = run(v, True)
        torch.cuda.synchronize()
        if rc != 0:
            say("  set %d: rc %d %s" % (v, rc, _lib.last_error() if hasattr(_lib, "last_error") else ""))
            continue
        same = torch.equal(o, ref)
        sdiff = ((ws[:2 * cout] - ref_stats).abs() / ref_stats.abs().clamp_min(1e-6)).max().item()
        o.zero_()
        run(v, False)
        torch.cuda.synchronize()
        same_plain = torch.equal(o, ref)
        t = timed(lambda: run(v, False))
        ts = timed(lambda: run(v, True))
        say("  set %d: %7.1f us (%6.0f GB/s alg, %.3f of 8 TB/s)  with statistics %7.1f us   rows identical %s / %s   "
            "statistics max rel diff %.1e" % (v, t, balg / t / 1e3, balg / t / 8e6, ts, same, same_plain, sdiff))
L.d3d_set_tuning(11, -1)
