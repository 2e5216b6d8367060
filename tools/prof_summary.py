"""Summarise rocprofv3 rocpd sqlite output on the GPU box (the .db files exceed what gpurun
copies back).  usage:
    prof_summary.py kernels <db> <window_ms> <steps> <out.txt> [header...]
    prof_summary.py gaps <db> <window_ms> <steps> <out.txt>      (appends)
    prof_summary.py pmc <db> <counter> <out.txt>
    prof_summary.py timeline <db> <window_ms> <out.txt>
    prof_summary.py roofjson <db> <window_ms> <steps> <out.json> <workload> <dtype>
"""
import sqlite3
import sys


def kernels(db, window_ms, steps, out, header):
    cur = sqlite3.connect(db).cursor()
    t1 = cur.execute("select max(end) from kernels").fetchone()[0]
    w0 = t1 - int(float(window_ms) * 1e6)
    rows = cur.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3 from kernels "
                       "where start > ? group by name order by 3 desc", (w0,)).fetchall()
    tot = sum(r[2] for r in rows)
    n = sum(r[1] for r in rows)
    steps = float(steps)
    with open(out, "w") as f:
        for h in header:
            f.write("# %s\n" % h)
        f.write("# steady-state window = last %.1f ms of the trace (%g timed steps)\n" % (float(window_ms), steps))
        f.write("# %d kernels, GPU busy %.2f ms/step, %d launches/step\n" % (n, tot / 1e3 / steps, n / steps))
        f.write("%12s %12s %10s %7s  %s\n" % ("us/step", "calls/step", "avg_us", "pct", "kernel"))
        for r in rows[:80]:
            f.write("%12.1f %12.1f %10.2f %6.1f%%  %s\n" % (r[2] / steps, r[1] / steps, r[3], 100 * r[2] / tot, r[0][:130]))


def gaps(db, window_ms, steps, out):
    """idle time on the device between consecutive kernels, attributed to the kernel that FOLLOWS the gap
    and to the one that precedes it -- shows which launches the stream waits for."""
    cur = sqlite3.connect(db).cursor()
    t1 = cur.execute("select max(end) from kernels").fetchone()[0]
    w0 = t1 - int(float(window_ms) * 1e6)
    rows = cur.execute("select name, start, end from kernels where start > ? order by start", (w0,)).fetchall()
    steps = float(steps)
    after, before = {}, {}
    tot = 0.0
    for (n0, s0, e0), (n1, s1, e1) in zip(rows[:-1], rows[1:]):
        g = max(0, s1 - e0) / 1e3
        tot += g
        a = after.setdefault(n1[:90], [0, 0.0])
        a[0] += 1
        a[1] += g
        b = before.setdefault(n0[:90], [0, 0.0])
        b[0] += 1
        b[1] += g
    busy = sum(e - s for _, s, e in rows) / 1e3
    with open(out, "a") as f:
        f.write("\n# gaps: window %.1f ms, busy %.2f ms/step, idle between kernels %.2f ms/step\n"
                % (float(window_ms), busy / 1e3 / steps, tot / 1e3 / steps))
        for title, d in (("gap BEFORE this kernel (it waited)", after), ("gap AFTER this kernel", before)):
            f.write("# %s\n%12s %10s %10s  kernel\n" % (title, "us/step", "n/step", "avg_us"))
            for k, (c, g) in sorted(d.items(), key=lambda kv: -kv[1][1])[:25]:
                f.write("%12.1f %10.1f %10.2f  %s\n" % (g / steps, c / steps, g / c, k))


def timeline(db, window_ms, out):
    """every kernel of the last `window_ms` of the trace, in start order: start (us from the window's first kernel),
    duration, queue / stream ids when the view has them, name -- to see WHERE a step waits, not just how long in total"""
    cur = sqlite3.connect(db).cursor()
    cols = [d[0] for d in cur.execute("select * from kernels limit 0").description]
    extra = [c for c in ("queue_id", "stream_id", "tid") if c in cols]
    t1 = cur.execute("select max(end) from kernels").fetchone()[0]
    w0 = t1 - int(float(window_ms) * 1e6)
    rows = cur.execute("select start, end, name%s from kernels where start > ? order by start"
                       % "".join(", " + c for c in extra), (w0,)).fetchall()
    with open(out, "w") as f:
        f.write("# columns of the kernels view: %s\n" % " ".join(cols))
        f.write("# %10s %9s %9s %s name\n" % ("start_us", "dur_us", "gap_us", " ".join(extra)))
        t0 = rows[0][0] if rows else 0
        last_end = t0
        for r in rows:
            f.write("%12.1f %9.1f %9.1f %s %s\n" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, (r[0] - last_end) / 1e3,
                                                   " ".join(str(v) for v in r[3:]), r[2][:70]))
            last_end = max(last_end, r[1])


def pmc(db, counter, out):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), avg(counter_value), sum(counter_value), avg(duration)/1e3 from pmc_events "
                       "where counter_name = ? group by name order by 4 desc", (counter,)).fetchall()
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --pmc %s ; per kernel: launches, avg counter value per launch, avg duration us\n" % counter)
        f.write("# (counter summed over its instances; FETCH_SIZE/WRITE_SIZE are in KiB -- see MI355X_MICROARCH.md HBM notes)\n")
        ours = ("spconv", "rb_", "vox_", "bn_", "pfn_", "dense_move", "nms_", "rotate_iou", "d3d_scan")
        keep = [r for i, r in enumerate(rows) if i < 40 or any(t in r[0] for t in ours)]
        for r in keep:
            f.write("%8d %16.1f %10.2f  %s\n" % (r[1], r[2], r[4], r[0][:130]))


def roofjson(db, window_ms, steps, out, workload, dtype):
    """the roofline kernel's launch durations in the steady-state window of a kernel trace -> a small json bench.py
    attaches to its `roofline.sources` (stamped with tools/stamp.py's source sha)"""
    import json
    import os
    import re

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import stamp

    cur = sqlite3.connect(db).cursor()
    t1 = cur.execute("select max(end) from kernels").fetchone()[0]
    w0 = t1 - int(float(window_ms) * 1e6)
    pat = r"spconv_fwd_mfma_kernel<64, 64, \d+, \d+, \d+, false, (false|true)>" if dtype == "fp32" else \
        r"spconv_fwd_(group_kernel<64, 64, \d+, \d+, \d+>|mfma_kernel<64, 64, \d+, \d+, \d+, true, (false|true)>)"
    rows = cur.execute("select name, count(*), sum(end-start)/1e3 from kernels where start > ? group by name", (w0,)).fetchall()
    mine = [(n, c, us) for n, c, us in rows if re.search(pat, n)]
    launches = sum(c for _, c, _ in mine)
    total_us = sum(us for _, _, us in mine)
    # launches that had the device to themselves: no weight-gradient kernel (the second stream of the backward pass,
    # sparse_stack.WGRAD_STREAM) running at any time between their start and end
    allk = cur.execute("select name, start, end from kernels where start > ? order by start", (w0,)).fetchall()
    wg = sorted((a, b) for n, a, b in allk if "wgrad" in n)
    import bisect
    starts = [a for a, _ in wg]
    solo_n, solo_us = 0, 0.0
    for n, a, b in allk:
        if not re.search(pat, n):
            continue
        i = bisect.bisect_left(starts, b)  # intervals starting before this launch ends
        if any(e > a for _, e in wg[max(0, i - 8):i]):
            continue
        solo_n += 1
        solo_us += (b - a) / 1e3
    d = {"_comment": "rocprofv3 --kernel-trace --stats of `bench.py --steps N --warmup 5` (tools/profile_step.sh): every forward / "
                     "data-gradient launch of the 64->64 sparse gather kernel (plain and parity-class-planned variants) in the "
                     "steady-state window",
         "workload": workload, "dtype": dtype, "steps": float(steps), "source_sha16": stamp.source_sha16(),
         "launches_per_step": launches / float(steps), "avg_launch_us": round(total_us / launches, 3) if launches else None,
         "solo": {"launches_per_step": solo_n / float(steps), "avg_launch_us": round(solo_us / solo_n, 3) if solo_n else None,
                  "what": "launches with no weight-gradient kernel (second stream) running between their start and end"},
         "variants": {n[:100]: {"launches": c, "avg_us": round(us / c, 3)} for n, c, us in mine}}
    json.dump(d, open(out, "w"), indent=1)
    print("wrote", out, d["avg_launch_us"])


if __name__ == "__main__":
    if sys.argv[1] == "roofjson":
        roofjson(*sys.argv[2:8])
    elif sys.argv[1] == "kernels":
        kernels(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6:])
    elif sys.argv[1] == "gaps":
        gaps(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5])
    elif sys.argv[1] == "timeline":
        timeline(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
