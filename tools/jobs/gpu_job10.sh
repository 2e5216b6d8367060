#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2j10; mkdir -p $O
export TMPDIR=/tmp
echo "== failing tests"; timeout 900 python -m pytest tests/test_dist_gpu.py tests/test_graph_tail_gpu.py -q -m gpu --tb=long > $O/tests.log 2>&1; tail -5 $O/tests.log; grep -n "Error\|assert " $O/tests.log | head -20
echo "== pmc fwd"; timeout 1500 bash tools/pmc_fwd.sh $O/pmc_fwd.txt > /dev/null 2>&1; cat $O/pmc_fwd.txt | cut -c1-200
rocprofv3-avail list 2>/dev/null | grep -o "\b\(TA\|TCP\|TCC\|TD\|GRBM\|SPI\)_[A-Za-z0-9_]*" | sort -u | tr '\n' ' ' > $O/counters.txt; wc -c $O/counters.txt
