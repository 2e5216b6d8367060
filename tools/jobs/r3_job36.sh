#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for i in 1 2 3; do
timeout 200 python bench.py --gpus 1 --steps 60 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32', d['value'], d['ms_per_step'], d['config']['host_ms_per_step'], d['roofline']['frac'])"
timeout 200 python bench.py --gpus 1 --steps 60 --warmup 5 --no-cpu-baseline --no-extras --dtype bf16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16', d['value'], d['ms_per_step'], d['config']['host_ms_per_step'], d['roofline']['frac'])"
done
