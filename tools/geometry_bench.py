"""The geometry stage ALONE (voxelizer + every rulebook of the backbone + row plans), nothing else on the GPU: isolated
kernel durations for rocprofv3 (tools/gpu_job.sh geoprof) and wall / event time per build.

    python tools/geometry_bench.py [workload] [chain 0|1] [reps]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import argparse

    import bench
    from det3d_amd import spconv

    name = sys.argv[1] if len(sys.argv) > 1 else "c2-20k"
    chain = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    spconv.RB_CHAIN = bool(chain)
    dev = torch.device("cuda", 0)
    args = argparse.Namespace(no_channels_last=False, cached_targets=False, augment=False, syncbn=False)
    wl = bench.Workload(name, args, dev, 0, 1, "fp32", 10)
    bb = wl.model.backbone

    def build(i):
        ex = wl.example(i, targets=False)
        if hasattr(bb, "plan_geometry"):
            ex["sparse_plan"] = bb.plan_geometry(ex["coordinates"], len(ex["num_voxels"]), ex["shape"][0])
        return ex

    for i in range(5):
        build(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(reps):
        build(i)
    e1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    print("%s chain=%d: %.3f ms wall, %.3f ms stream time per geometry build (voxelize + rulebooks + plans, 2 host syncs)" % (
        name, chain, 1e3 * wall, e0.elapsed_time(e1) / reps))


if __name__ == "__main__":
    main()
