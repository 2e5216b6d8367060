#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2j24; mkdir -p $O
export TMPDIR=/tmp
for w in c3 c5; do
  timeout 400 python bench.py --workload $w --steps 6 --warmup 2 --burn-in 0.3 --no-cpu-baseline --no-extras > $O/bench_$w.json 2> $O/bench_$w.err
  echo "$w: $(python tools/show_bench.py $O/bench_$w.json 2>&1 | head -1)"; tail -2 $O/bench_$w.err | cut -c1-300
done
