#!/bin/bash
# round-2 consolidation: whole GPU suite, smoke, driver-style bench, kernel profile, PMC passes
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2j6; mkdir -p $O
export TMPDIR=/tmp
echo "== full gpu suite"; timeout 2400 python -m pytest tests -q -m gpu --tb=short 2>&1 | tail -25 > $O/t_all.log; tail -6 $O/t_all.log
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
echo "== bench (driver flags)"; timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python tools/show_bench.py $O/bench_driver.json workloads; tail -4 $O/bench_driver.err
echo "== bench (defaults, 300 steps)"; timeout 1500 python bench.py --no-cpu-baseline --no-extras > $O/bench_300.json 2> /dev/null; python tools/show_bench.py $O/bench_300.json | head -1
echo "== profile"; timeout 900 bash tools/profile_step.sh $O/kernel_stats.txt; head -30 $O/kernel_stats.txt | cut -c1-150
echo "== pmc"; timeout 1500 bash tools/pmc_spconv.sh $O 2>&1 | tail -45
echo "== counters available"; rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' | head -c 3000
