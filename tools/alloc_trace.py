"""Which call sites still reach hipMalloc in the steady-state training step?  

Runs the threaded-prefetch loop of bench.py on the KITTI workload with the caching allocator's history recorder on and
prints the Python stacks of every `segment_alloc` event after the warm-up.  python tools/alloc_trace.py [warm] [steps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from det3d_amd import Config, build_detector, synthetic
    from det3d_amd.prefetch import GeometryPrefetcher
    from det3d_amd.trainer import TrainStep

    warm = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    cfg = Config.fromfile(os.path.join(ROOT, "det3d_amd", "configs", "kitti_car_second.py"))
    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    model = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(dev).train()
    model.backbone.dense_channels_last = True
    ds = synthetic.KittiCarSynthetic(n_points=20000, n_seeds=16 * batch)
    batches = [ds.device_batch(i, batch, dev) for i in range(16)]
    step = TrainStep(model, total_steps=1000)
    make = lambda i: synthetic.make_example(batches[i % 16], synthetic.KITTI_CAR, fused_mean=True, device_targets=True)
    pf = GeometryPrefetcher(make, model, dev)
    ex, nxt = pf.prepare(0), pf.submit(1)
    for i in range(warm + steps):
        if i == warm:
            torch.cuda.synchronize()
            torch.cuda.memory._record_memory_history(max_entries=200000)
            m0 = torch.cuda.memory_stats(dev)
        step(ex)
        ex = nxt.result()
        nxt = pf.submit(i + 2)
    torch.cuda.synchronize()
    m1 = torch.cuda.memory_stats(dev)
    snap = torch.cuda.memory._snapshot()
    torch.cuda.memory._record_memory_history(enabled=None)
    print("device allocs in %d steps after %d warm-up steps: %d, reserved %d -> %d MB" % (
        steps, warm, m1["num_device_alloc"] - m0["num_device_alloc"], m0["reserved_bytes.all.current"] >> 20,
        m1["reserved_bytes.all.current"] >> 20))
    n = 0
    for tr in snap.get("device_traces", []):
        for j, e in enumerate(tr):
            if e.get("action") != "segment_alloc":
                continue
            n += 1
            # the request that made the allocator grow is the `alloc` event right after the segment_alloc
            req = next((x for x in tr[j + 1: j + 4] if x.get("action") == "alloc"), e)
            frames = [f for f in req.get("frames", []) if "/det3d_amd/" in f.get("filename", "") or "bench" in f.get("filename", "")
                      or "alloc_trace" in f.get("filename", "")]
            print("segment_alloc %8.2f MB (request %8.3f MB) stream %s" % (e["size"] / 2**20, req.get("size", 0) / 2**20, e.get("stream")))
            for f in frames[:6]:
                print("      %s:%d %s" % (os.path.relpath(f["filename"], ROOT), f["line"], f["name"]))
    print("%d segment_alloc events" % n)
    pf.close()


if __name__ == "__main__":
    main()
