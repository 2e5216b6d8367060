#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for t in 0 1 2 3 0 1 2 3; do
D3D_SMALL_TILE=$t timeout 300 python bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32 small=$t', d['value'], d['ms_per_step'], d['sparse_conv_ms_per_step'], {k[-7:]: v['avg_us'] for k, v in d['kernels'].items() if 'fwd' in k})"
done
