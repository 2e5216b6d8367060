"""compact view of a bench.py JSON line: tools/show_bench.py <file> [workloads]"""
import json
import sys

d = json.loads([ln for ln in open(sys.argv[1]).read().splitlines() if ln.startswith("{")][-1])
r = d.get("roofline", {})
print("value %.1f %s  %.3f ms/step  roofline %s (%s us, %s GB/s)  sparse %s ms" % (
    d["value"], d["unit"], d["ms_per_step"], r.get("frac"), r.get("avg_launch_us"), r.get("achieved"),
    d.get("sparse_conv_ms_per_step")))
c = d.get("config", {})
print("   host ms/step %s  graphed_tail %s  prefetch %s  burn-in %s" % (
    c.get("host_ms_per_step"), c.get("graphed_tail"), c.get("geometry_prefetch"), c.get("burn_in_steps")))
for k, v in (d.get("kernels") or {}).items():
    print("   %-34s x%-2d %7.2f us  %7.1f GB/s" % (k, v["launches_per_step"], v["avg_us"], v.get("achieved_GBs", 0)))
for k in ("fp32", "bf16", "predict", "rpn_mfma", "precision", "stages", "probe_error", "other_dtype_error"):
    if k in d:
        print("  ", k, json.dumps(d[k])[:400])
if len(sys.argv) > 2 and "workloads" in d:
    for k, v in d["workloads"].items():
        print("  ", k, json.dumps(v)[:700])
