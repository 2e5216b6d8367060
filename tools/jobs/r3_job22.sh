#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_spconv_gpu.py -q -m gpu --tb=short -x -k "planned" 2>&1 | tail -2
for t in 0 1 2 3 0 1 2 3; do
D3D_PLAN_TILE=$t timeout 300 python bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32 tile=$t', d['value'], d['ms_per_step'], d['sparse_conv_ms_per_step'], {k[-7:]: v['avg_us'] for k, v in d['kernels'].items() if 'fwd' in k})"
done
for t in 0 1 2 3; do
D3D_PLAN_TILE=$t timeout 300 python bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-extras --dtype bf16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 tile=$t', d['value'], d['ms_per_step'], d['sparse_conv_ms_per_step'], {k[-7:]: v['avg_us'] for k, v in d['kernels'].items() if 'fwd' in k})"
done
