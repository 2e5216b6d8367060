"""d3d_sparse_conv_fwd_pairs (pair-compacted kernel, round 6) against d3d_sparse_conv_fwd_ex on the rulebooks of the benched
KITTI batch (4 frames x 20k points through SpMiddleFHD's geometry), f32 features: every forward and data-gradient launch
shape of the backbone that has a pair-compacted tiling, alone on the device, HIP events, one process.

    python tools/bench_pairs.py [out.txt] [--workload kitti|kitti60k]

Per case: microseconds of both kernels (plain and with the BatchNorm-statistics epilogue), SURVEY 8(d) algorithmic GB/s
and fraction of 8 TB/s, max difference between the two kernels' rows relative to the largest row entry, tiles issued, and
the microseconds of the table build."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from det3d_amd import _lib, spconv, synthetic  # noqa: E402
from det3d_amd.point_cloud_ops import voxelize_batch  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
out = open(args[0], "w") if args else None
points = 60000 if "kitti60k" in sys.argv else 20000


def say(msg):
    print(msg, flush=True)
    if out:
        out.write(msg + "\n")
        out.flush()


dev = torch.device("cuda", 0)
ds = synthetic.KittiCarSynthetic(n_points=points, n_seeds=4)
b = ds.device_batch(0, 4, dev)
vb = voxelize_batch(b["points"], b["frame_offsets"], synthetic.KITTI_CAR["voxel_size"], synthetic.KITTI_CAR["pc_range"],
                    5, 20000, want_voxels=False, mean_channels=4).sync()
x = spconv.SparseConvTensor(vb["mean"], vb["coordinates"], [41, 1600, 1408], 4)
net = spconv.SparseSequential(
    spconv.SubMConv3d(4, 16, 3, bias=False, indice_key="subm0"),
    spconv.SparseConv3d(16, 32, 3, 2, padding=1, bias=False, indice_key="down1"),
    spconv.SubMConv3d(32, 32, 3, bias=False, indice_key="subm1"),
    spconv.SparseConv3d(32, 64, 3, 2, padding=1, bias=False, indice_key="down2"),
    spconv.SubMConv3d(64, 64, 3, bias=False, indice_key="subm2"),
    spconv.SparseConv3d(64, 64, 3, 2, padding=[0, 1, 1], bias=False, indice_key="down3"),
    spconv.SubMConv3d(64, 64, 3, bias=False, indice_key="subm3"),
    spconv.SparseConv3d(64, 64, (3, 1, 1), (2, 1, 1), bias=False, indice_key="down4")).to(dev)
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


# (cin, cout, rulebook key, side): forward launches ("out") and the strided layers' data gradients ("in": gradient
# channels -> input channels over the input-stationary table)
cases = [(32, 32, "subm1", "out"), (32, 64, "down2", "out"), (64, 32, "down2", "in"), (64, 64, "subm2", "out"),
         (64, 64, "down3", "out"), (64, 64, "down3", "in"), (64, 64, "subm3", "out"), (64, 64, "down4", "out"),
         (64, 64, "down4", "in")]
tot_old = tot_new = 0.0
for cin, cout, key, side in cases:
    rb = x.indice_dict[key]
    kvol = rb.kvol
    nbr, cap, n, n_src = (rb.nbr_out, rb.cap_out, rb.n_out, rb.n_in) if side == "out" else (rb.nbr_in, rb.cap_in, rb.n_in, rb.n_out)
    tab2d = nbr.view(kvol, cap)[:, :n]
    P = int((tab2d >= 0).sum().item())
    f = torch.randn(n_src, cin, device=dev)
    w32 = torch.randn(kvol, cin, cout, device=dev) * 0.05
    wpack = torch.empty((kvol, cin, cout), device=dev)
    _lib.check(L.d3d_pack_conv_weights(w32.data_ptr(), kvol, cin, cout, _lib.D3D_F32, 1, wpack.data_ptr(), None, st), "pack")
    nbytes = L.d3d_pair_table_bytes(n, kvol)
    table = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    build = lambda: L.d3d_pair_table_build(nbr.data_ptr(), cap, n, kvol, table.data_ptr(), nbytes, st)
    _lib.check(build(), "build")
    t_build = timed(build, 20)
    cap64 = (n + 63) // 64 * 64
    off = (kvol * cap64 * 4 + 255) // 256 * 256
    cnt = table[off: off + cap64 // 64 * 32].view(cap64 // 64, 32)[:, :kvol].int()
    tiles = int(((cnt + 15) // 16).sum().item())
    act16 = int((tab2d[:, : n // 16 * 16].reshape(kvol, n // 16, 16) >= 0).any(2).sum().item())
    o_new = torch.empty(n, cout, device=dev)
    o_old = torch.empty(n, cout, device=dev)
    nws = L.d3d_conv_epilogue_ws_floats(n, cout)
    ws = torch.zeros(nws, device=dev)
    balg = P * cin * 4 + P * 8 + n * cout * 4 + kvol * cin * cout * 4

    def new(stats=False):
        ep = _lib.ConvEpilogue(stats=ws.data_ptr(), ws_floats=nws) if stats else None
        return L.d3d_sparse_conv_fwd_pairs(f.data_ptr(), wpack.data_ptr(), None, table.data_ptr(), n, kvol, cin, cout,
                                           _lib.D3D_F32, o_new.data_ptr(), ctypes.byref(ep) if ep else None, st)

    def old(stats=False):
        ep = _lib.ConvEpilogue(stats=ws.data_ptr(), ws_floats=nws) if stats else None
        return L.d3d_sparse_conv_fwd_ex(f.data_ptr(), wpack.data_ptr(), None, nbr.data_ptr(), cap, n, kvol, cin, cout,
                                        _lib.D3D_F32, o_old.data_ptr(), ctypes.byref(ep) if ep else None, st)

    _lib.check(new(), "pairs")
    _lib.check(old(), "tile")
    torch.cuda.synchronize()
    diff = float((o_new - o_old).abs().max()) / float(o_old.abs().max())
    tn, to = timed(new), timed(old)
    tns, tos = timed(lambda: new(True)), timed(lambda: old(True))
    tot_old += to
    tot_new += tn
    say("%2d -> %2d %-5s %-3s n = %6d  pairs %7d (%5.2f/row)  tiles %6d (16-row tiles with a pair: %6d, x%.2f)  build %5.1f us | "
        "pairs %6.1f us (%.3f of 8 TB/s; stats %6.1f) | tile kernel %6.1f us (%.3f; stats %6.1f) | x%.2f | rel diff %.1e"
        % (cin, cout, key, side, n, P, P / n, tiles, act16, act16 / max(tiles, 1), t_build, tn, balg / tn / 8e6, tns, to,
           balg / to / 8e6, tos, to / tn, diff))
say("sum over the cases: tile kernels %.1f us, pair kernels %.1f us (x%.2f)" % (tot_old, tot_new, tot_old / tot_new))
