"""f32 (split arithmetic) tilings of the 128-channel sparse layers of the CBGS backbones (SpMiddleResNetFHD,
det3d/models/backbones/scn.py:308-370) on a nuScenes-scale rulebook: d3d_set_tuning knob 0, variants 60..63
(csrc/d3d_spconv.hip), each checked bit for bit against the default tiling, HIP events.  python tools/tune_fwd_wide.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from det3d_amd import Config, _lib, spconv, synthetic  # noqa: E402

dev = torch.device("cuda", 0)
cfg = Config.fromfile(os.path.join(ROOT, "det3d_amd", "configs", "nusc_cbgs.py"))
ds = synthetic.TaskGroupSynthetic(cfg, 250000, n_seeds=8, beams=32, sweeps=10)
ex = ds.make_example(ds.device_batch(0, 8, dev), targets=False)
shape = [int(s) for s in ex["shape"][0]]
x = spconv.SparseConvTensor(ex["voxel_mean"], ex["coordinates"], [shape[2] + 1, shape[1], shape[0]], 8)
# the geometry of the backbone's last two stages: three strided layers down to the 128-channel grid, then its
# submanifold rulebook (subm) and the strided rulebook into it (down)
net = spconv.SparseSequential(spconv.SparseConv3d(5, 16, 3, 2, padding=1, bias=False),
                              spconv.SparseConv3d(16, 32, 3, 2, padding=1, bias=False),
                              spconv.SparseConv3d(32, 64, 3, 2, padding=[0, 1, 1], bias=False, indice_key="down"),
                              spconv.SubMConv3d(64, 64, 3, bias=False, indice_key="subm")).to(dev)
net.plan(x)
L = _lib.lib()
st = _lib.stream_ptr(dev)


def timed(call, reps=20):
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


names = {0: "default MT1 x16w D1", 60: "MT2 x8w D1", 61: "MT1 x8w D2", 62: "MT1 x16w D2", 63: "MT2 x8w D2"}
for key, cin, cout in (("subm", 128, 128), ("subm", 64, 128), ("subm", 128, 64), ("down", 64, 128)):
    rb = x.indice_dict[key]
    n_in, n, P = rb.n_in, rb.n_out, rb.pairs()
    f = torch.randn(n_in, cin, device=dev)
    w32 = torch.randn(rb.kvol, cin, cout, device=dev) * 0.05
    wpack = torch.empty((2 * rb.kvol, cout, cin), device=dev, dtype=torch.bfloat16)  # hi / lo planes
    _lib.check(L.d3d_pack_conv_weights(w32.data_ptr(), rb.kvol, cin, cout, _lib.D3D_F32, int(rb.subm), wpack.data_ptr(), None, st), "pack")
    o = torch.empty(n, cout, device=dev)
    balg = P * cin * 4 + P * 8 + n * cout * 4 + rb.kvol * cin * cout * 4

    def run(v):
        L.d3d_set_tuning(0, v)
        return L.d3d_sparse_conv_fwd_ex(f.data_ptr(), wpack.data_ptr(), None, rb.nbr_out.data_ptr(), rb.cap_out, n, rb.kvol,
                                        cin, cout, _lib.D3D_F32, o.data_ptr(), None, st)

    print("forward %s %d->%d f32 (split): %d -> %d sites, %d pairs, B_alg %.1f MB" % (key, cin, cout, n_in, n, P, balg / 1e6),
          flush=True)
    run(0)
    torch.cuda.synchronize()
    ref = o.clone()
    for v in (0, 60, 61, 62, 63):
        if v == 63 and cin == cout:
            continue
        o.zero_()
        rc = run(v)
        torch.cuda.synchronize()
        same = torch.equal(o, ref)
        t = timed(lambda: run(v))
        print("  variant %2d %-20s rc %d  %7.1f us  %6.1f GB/s alg = %.3f of 8 TB/s  identical %s" % (
            v, names[v], rc, t, balg / t / 1e3, balg / t / 1e3 / 8000, same), flush=True)
    L.d3d_set_tuning(0, 0)

# ---- weight gradient of the same layers: target block count (d3d_set_tuning knob 1; default one block per CU).  The kernel
# launches (offset groups x splits) blocks with splits rounded to a multiple of 8: 128x128 (9 groups) gets 216 blocks at
# the default -- 40 CUs idle
print("\nweight gradient, f32 (split): us per launch by target block count", flush=True)
for key, cin, cout in (("subm", 128, 128), ("subm", 64, 64), ("subm", 32, 32), ("down", 64, 128)):
    rb = x.indice_dict[key]
    n_in, n, P = rb.n_in, rb.n_out, rb.pairs()
    f = torch.randn(n_in, cin, device=dev)
    g = torch.randn(n, cout, device=dev)
    dw = torch.empty((rb.kvol, cin, cout), device=dev)
    row = []
    ref = None
    for blocks in (0, 288, 384, 512, 768, 1024):
        L.d3d_set_tuning(1, blocks)
        wsb = L.d3d_sparse_conv_wgrad_workspace_bytes(n, rb.kvol, cin, cout, _lib.D3D_F32)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
        call = lambda: L.d3d_sparse_conv_wgrad(f.data_ptr(), g.data_ptr(), rb.nbr_out.data_ptr(), rb.cap_out, n, rb.kvol, cin,
                                               cout, _lib.D3D_F32, dw.data_ptr(), ws.data_ptr(), wsb, st)
        rc = call()
        torch.cuda.synchronize()
        if ref is None:
            ref = dw.clone()
        close = float((dw - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
        row.append("%s %.0f%s" % ("default" if blocks == 0 else blocks, timed(call, 10), "" if (rc == 0 and close) else " (rc %d, mismatch)" % rc))
    L.d3d_set_tuning(1, 0)
    print("  %s %d x %d (%d pairs, B_alg %.0f MB): %s" % (key, cin, cout, P, P * (cin + cout) * 4 / 1e6, " | ".join(row)), flush=True)

# ---- dout fragments in registers (knob 1 value = 100000 * variant + blocks): 0 = default (f32: in registers, bf16: re-read
# from LDS), 1 = re-read every stage (round 3), 2 = in registers at the round-3 prefetch depth, 3 = 8-wave blocks (another
# split of the sites: compared with a tolerance instead of bit for bit)
print("\nweight gradient: us per launch by kernel variant (0 default | 1 dout re-read per stage | 2 dout fragments in registers | 3 8-wave blocks)", flush=True)
for tdt, ddt, name in ((torch.float32, _lib.D3D_F32, "f32"), (torch.bfloat16, _lib.D3D_BF16, "bf16")):
    for key, cin, cout in (("subm", 128, 128), ("subm", 64, 64), ("subm", 32, 32), ("subm", 16, 16), ("down", 64, 128), ("down", 32, 64)):
        rb = x.indice_dict[key]
        n_in, n, P = rb.n_in, rb.n_out, rb.pairs()
        f = torch.randn(n_in, cin, device=dev).to(tdt)
        g = torch.randn(n, cout, device=dev).to(tdt)
        dw = torch.empty((rb.kvol, cin, cout), device=dev)
        row = []
        ref = None
        for variant in (0, 1, 2, 3):
            L.d3d_set_tuning(1, 100000 * variant)
            wsb = L.d3d_sparse_conv_wgrad_workspace_bytes(n, rb.kvol, cin, cout, ddt)
            ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
            call = lambda: L.d3d_sparse_conv_wgrad(f.data_ptr(), g.data_ptr(), rb.nbr_out.data_ptr(), rb.cap_out, n, rb.kvol, cin,
                                                   cout, ddt, dw.data_ptr(), ws.data_ptr(), wsb, st)
            rc = call()
            torch.cuda.synchronize()
            if ref is None:
                ref = dw.clone()
            same = torch.equal(dw, ref) if variant != 3 else float((dw - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
            row.append("%d: %.0f%s" % (variant, timed(call, 10), "" if (rc == 0 and same) else " (rc %d, DIFFERENT)" % rc))
        L.d3d_set_tuning(1, 0)
        print("  %s %s %d x %d (%d pairs): %s" % (name, key, cin, cout, P, " | ".join(row)), flush=True)
