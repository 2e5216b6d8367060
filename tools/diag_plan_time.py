"""host and GPU time of SparseSequential.plan (all rulebooks of the backbone) on the current stream, with and without
the row plans of the strided data gradients:  python tools/diag_plan_time.py"""
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from det3d_amd import spconv  # noqa: E402

args = types.SimpleNamespace(no_channels_last=False, cached_targets=False, augment=False)
dev = torch.device("cuda", 0)
wl = bench.Workload("c2-20k", args, dev, 0, 1, "fp32", 100)
ex = wl.example(0, targets=True)
shape = [int(s) for s in ex["shape"][0]]
st = spconv.SparseConvTensor(ex["voxel_mean"], ex["coordinates"], [shape[2] + 1, shape[1], shape[0]], wl.batch)
bb = wl.model.backbone
for plan in (1, 0, 1, 0):
    spconv.DGRAD_PLAN = bool(plan)
    for rep in range(4):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        d = bb.middle_conv.plan(st)
        e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print("DGRAD_PLAN=%d rep %d: host %.3f ms, stream %.3f ms" % (plan, rep, (t1 - t0) * 1e3, e0.elapsed_time(e1)), flush=True)
