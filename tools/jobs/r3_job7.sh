#!/bin/bash
# round 3, job 7: checkpoint -- the full GPU suite and the bench exactly as the driver runs it
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3j7
mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu --tb=short -x 2>&1 | tail -30 > $O/tests.log
tail -5 $O/tests.log
/usr/bin/time -v timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
grep -E "Elapsed|Maximum resident" $O/bench.err
python tools/show_bench.py $O/bench.json workloads | cut -c1-600
