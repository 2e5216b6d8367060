#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2j22; mkdir -p $O
export TMPDIR=/tmp
echo "== bench (driver flags)"; timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python tools/show_bench.py $O/bench_driver.json all | cut -c1-300 | head -24
echo "== bench (300 steps, no extras)"; timeout 900 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-extras > $O/bench_300.json 2> /dev/null; python tools/show_bench.py $O/bench_300.json | head -2
echo "== profile"; timeout 900 bash tools/profile_step.sh $O/kernel_stats.txt; head -16 $O/kernel_stats.txt | cut -c1-150
