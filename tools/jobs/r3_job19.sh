#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_spconv_gpu.py tests/test_determinism_gpu.py -q -m gpu --tb=short -x 2>&1 | tail -3
for i in 1 2; do
timeout 300 python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-launch-log 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32', d['value'], d['ms_per_step'], d['config']['host_ms_per_step'])"
timeout 300 python bench.py --gpus 1 --steps 40 --warmup 5 --dtype bf16 --no-cpu-baseline --no-extras --no-launch-log 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16', d['value'], d['ms_per_step'], d['config']['host_ms_per_step'])"
done
D3D_DENSE_STATS_EPILOGUE_BF16=0 timeout 300 python bench.py --gpus 1 --steps 40 --warmup 5 --dtype bf16 --no-cpu-baseline --no-extras --no-launch-log 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 no-epi', d['value'], d['ms_per_step'], d['config']['host_ms_per_step'])"
