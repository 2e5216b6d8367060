"""loss trajectory of a bench workload at a given batch size / precision (diagnostic):
    python tools/diag_workload.py c4 16 fp32 [steps]"""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

name, batch, dtype = sys.argv[1], int(sys.argv[2]), sys.argv[3]
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 8
args = types.SimpleNamespace(no_channels_last=False, cached_targets=False)
dev = torch.device("cuda", 0)
wl = bench.Workload(name, args, dev, 0, 1, dtype, 100, batch=batch)
for i in range(steps):
    ex = wl.example(i)
    loss = wl.step_fn(ex)
    torch.cuda.synchronize()
    print("%s batch %d %s step %d: loss %.4f grad_norm %.4f voxels %d" % (
        name, batch, dtype, i, float(loss), float(wl.step_fn.last["grad_norm"]), int(ex["coordinates"].shape[0])), flush=True)
