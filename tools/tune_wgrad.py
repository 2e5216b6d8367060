"""sweep the wgrad launch knobs on a real KITTI-scale rulebook (run on the GPU box).
Each configuration runs in a child process because the knobs are read once per process."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch
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
    n = rb.n_out
    L = _lib.lib()
    for (ci, co) in ((64, 64), (32, 32)):
        f = torch.randn(n, ci, device=dev).bfloat16(); d = torch.randn(n, co, device=dev).bfloat16()
        dw = torch.empty(27, ci, co, device=dev)
        st = _lib.stream_ptr(dev)
        call = lambda: L.d3d_sparse_conv_wgrad(f.data_ptr(), d.data_ptr(), rb.nbr_out.data_ptr(), rb.cap_out, n, 27, ci, co, 1, dw.data_ptr(), st)
        for _ in range(3): call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): call()
        e1.record(); torch.cuda.synchronize()
        print("  n=%d pairs=%d %dx%d: %.1f us/launch" % (n, rb.pairs(), ci, co, e0.elapsed_time(e1) * 1000 / 20))
else:
    for kg in (3, 1):
        for variant in (0, 1, 2):
            if kg == 1 and variant == 2:
                continue
            for blocks in (1024, 2048, 4096):
                env = dict(os.environ, D3D_WGRAD_KG=str(kg), D3D_WGRAD_BLOCKS=str(blocks), D3D_WGRAD_VARIANT=str(variant))
                print("KG=%d variant=%d blocks=%d" % (kg, variant, blocks), flush=True)
                subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env)
