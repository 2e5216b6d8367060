#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_optim_gpu.py tests/test_e2e_gpu.py tests/test_pfn_gpu.py -q -m gpu --tb=short -x 2>&1 | tail -6
for a in "" "--no-channels-last"; do
timeout 300 python bench.py --gpus 1 --workload c3 --steps 10 --warmup 3 --burn-in 0.5 --no-cpu-baseline --no-extras $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 $a', d['value'], d['ms_per_step'], d.get('final_loss'))"
done
