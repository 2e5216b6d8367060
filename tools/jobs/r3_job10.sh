#!/bin/bash
# round 3, job 10: full GPU suite with the bit-equality comparisons; graphed tail A/B in both precisions
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3j10
mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu --tb=short 2>&1 | tail -40 > $O/tests.log
tail -8 $O/tests.log
for g in 0 1; do for dt in fp32 bf16; do
D3D_GRAPH_TAIL=$g timeout 300 python bench.py --gpus 1 --steps 40 --warmup 5 --dtype $dt --no-cpu-baseline --no-extras --no-launch-log 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('graph $g $dt', d['value'], d['ms_per_step'], d['config']['host_ms_per_step'], d['config']['graphed_tail'], d['config']['final_loss'])"
done; done | tee $O/graph_ab.txt
