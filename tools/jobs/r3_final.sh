#!/bin/bash
# the driver's round-end sequence: GPU suite (-x), smoke, bench at the driver's flags
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3final}
mkdir -p $O
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/tests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
if [ "$2" != "nobench" ]; then
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/round3_bench_driver_flags.json 2> $O/bench.err
timeout 300 python bench.py --gpus 1 --steps 300 --warmup 10 --no-cpu-baseline --no-extras > $O/round3_bench_300steps.json 2> $O/bench300.err; python tools/show_bench.py $O/round3_bench_300steps.json | head -2
python tools/show_bench.py $O/round3_bench_driver_flags.json | head -3
python -c "
import json; d=json.loads(open('$O/round3_bench_driver_flags.json').read().strip().splitlines()[-1]); print(d['roofline']['traffic'], d['roofline'].get('traffic_source'), d['stages']['rulebooks'])"
fi
