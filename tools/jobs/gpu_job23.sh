#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2j23; mkdir -p $O
export TMPDIR=/tmp
echo "== bench (driver flags)"; timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python tools/show_bench.py $O/bench_driver.json | head -2
echo "== bench (driver flags, no extras) x2"; for i in 1 2; do timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_d$i.json 2>/dev/null; python tools/show_bench.py $O/bench_d$i.json | head -1; done
