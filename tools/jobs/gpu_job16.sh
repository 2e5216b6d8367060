#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2j18; mkdir -p $O
export TMPDIR=/tmp
echo "== full gpu suite"; timeout 1500 python -m pytest tests -q -m gpu --tb=short > $O/tests.log 2>&1; tail -4 $O/tests.log; grep -n "^E  \|^FAILED" $O/tests.log | head -20
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== pmc"; timeout 1200 bash tools/pmc_spconv.sh $O > $O/pmc.log 2>&1; tail -3 $O/pmc.log; cp $O/round2_pmc_spconv.json profiles/round2_pmc_spconv.json 2>/dev/null
echo "== bench (driver flags)"; timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python tools/show_bench.py $O/bench_driver.json all | cut -c1-400; tail -3 $O/bench_driver.err
echo "== bench (300 steps, no extras)"; timeout 900 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-extras > $O/bench_300.json 2> /dev/null; python tools/show_bench.py $O/bench_300.json | head -2
echo "== profile"; timeout 900 bash tools/profile_step.sh $O/kernel_stats.txt; head -12 $O/kernel_stats.txt | cut -c1-150
