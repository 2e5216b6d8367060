"""sweep (MT, WAVES) tilings of spconv_fwd_mfma_kernel<64,64> on a real KITTI-scale rulebook."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from det3d_amd import spconv, synthetic, _lib
    from det3d_amd.point_cloud_ops import voxelize_batch
    dev = torch.device("cuda", 0)
    ds = synthetic.KittiCarSynthetic(n_points=20000, n_seeds=4)
    b = ds.device_batch(0, 4, dev)
    vb = voxelize_batch(b["points"], b["frame_offsets"], synthetic.KITTI_CAR["voxel_size"], synthetic.KITTI_CAR["pc_range"], 5, 20000, want_voxels=False, mean_channels=4).sync()
    x = spconv.SparseConvTensor(vb["mean"], vb["coordinates"], [41, 1600, 1408], 4)
    net = spconv.SparseSequential(spconv.SparseConv3d(4, 16, 3, 2, padding=1, bias=False), spconv.SparseConv3d(16, 64, 3, 2, padding=1, bias=False),
                                  spconv.SubMConv3d(64, 64, 3, bias=False, indice_key="s2")).to(dev)
    net.plan(x)
    rb = x.indice_dict["s2"]
    n = rb.n_out; P = rb.pairs()
    L = _lib.lib()
    f = torch.randn(n, 64, device=dev).bfloat16(); w = torch.randn(27, 64, 64, device=dev).bfloat16(); out = torch.empty(n, 64, device=dev, dtype=torch.bfloat16)
    st = _lib.stream_ptr(dev)
    call = lambda: L.d3d_sparse_conv_fwd(f.data_ptr(), w.data_ptr(), None, rb.nbr_out.data_ptr(), rb.cap_out, n, 27, 64, 64, 1, out.data_ptr(), st)
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 30
    balg = P * 64 * 2 + P * 8 + n * 64 * 2 + 27 * 64 * 64 * 2
    print("  n=%d pairs=%d: %.1f us/launch, %.0f GB/s algorithmic (%.1f%% of 8 TB/s)" % (n, P, us, balg / us / 1e3, balg / us / 1e3 / 80))
else:
    names = {0: "MT1 x 16 waves (default)", 1: "MT4 x 4", 2: "MT4 x 8", 3: "MT2 x 4", 4: "MT1 x 8", 5: "MT2 x 16", 6: "MT2 x 8"}
    for v in range(7):
        print("variant %d: %s" % (v, names[v]), flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, D3D_FWD_VARIANT=str(v)))
