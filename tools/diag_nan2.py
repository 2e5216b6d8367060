"""bisect the non-finite final loss of the bench loop (diagnostic)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

sys.argv = ["bench.py", "--steps", "30", "--warmup", "2", "--no-cpu-baseline", "--no-extras"]
args = bench.parse()
dev = torch.device("cuda", 0)
note = lambda m: None
for label, kw in [("burn0 nofreeze nosample", dict(burn_in_s=0.0, freeze=False, launch_filter=None)),
                  ("burn0 freeze nosample", dict(burn_in_s=0.0, freeze=True, launch_filter=None)),
                  ("burn0 nofreeze sample", dict(burn_in_s=0.0, freeze=False, launch_filter=lambda k, ci, co: k != "wgrad" and ci == 64 and co == 64)),
                  ("burn1.0 nofreeze nosample", dict(burn_in_s=1.0, freeze=False, launch_filter=None))]:
    wl = bench.Workload("c2-20k", args, dev, 0, 1, "bf16", 32)
    el, rec, loss = bench.timed_region(wl, 30, 2, 0, 1, dev, note, prefetch=True, **kw)
    print(label, "final loss", float(loss), "ms/step %.3f" % (el / 30 * 1e3), "gnorm", float(wl.step_fn.last["grad_norm"]), flush=True)
    del wl
    torch.cuda.empty_cache()
