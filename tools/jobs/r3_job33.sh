#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_pfn_gpu.py tests/test_e2e_gpu.py -q -m gpu --tb=short -x -k "pfn or pp or pillar" 2>&1 | tail -4
timeout 300 python bench.py --gpus 1 --workload c3 --steps 10 --warmup 3 --burn-in 0.5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3', d['value'], d['ms_per_step'])"
O=$GRAFT_REPO_ROOT/gpurun_out/r3j33; mkdir -p $O
bash tools/profile_step.sh $O/kernel_stats_c3.txt --workload c3 --burn-in 0 > $O/prof_c3.log 2>&1
grep -n "pfn_" $O/kernel_stats_c3.txt | head -8 | cut -c1-120
