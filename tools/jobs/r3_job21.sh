#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_spconv_gpu.py -q -m gpu --tb=short -x -k "plan or oracle" 2>&1 | tail -8
for w in 1 0 1 0; do
D3D_DGRAD_PLAN=$w timeout 300 python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32 plan=$w', d['value'], d['ms_per_step'], d['sparse_conv_ms_per_step'], {k: v['avg_us'] for k, v in d['kernels'].items() if 'fwd' in k})"
done
D3D_DGRAD_PLAN=1 timeout 300 python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras --dtype bf16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 plan=1', d['value'], d['ms_per_step'], d['sparse_conv_ms_per_step'], {k: v['avg_us'] for k, v in d['kernels'].items() if 'fwd' in k})"
D3D_DGRAD_PLAN=0 timeout 300 python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras --dtype bf16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 plan=0', d['value'], d['ms_per_step'], d['sparse_conv_ms_per_step'], {k: v['avg_us'] for k, v in d['kernels'].items() if 'fwd' in k})"
