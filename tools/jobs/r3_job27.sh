#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_optim_gpu.py -q -m gpu --tb=short -x 2>&1 | tail -5
timeout 300 python tools/diag_adam.py fp32 2>&1 | grep "losses"
for f in 1 0 1 0; do
D3D_FUSED_ADAM=$f timeout 300 python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32 fused_adam=$f', d['value'], d['ms_per_step'], d['config']['host_ms_per_step'], d.get('final_loss'))"
done
for f in 1 0; do
D3D_FUSED_ADAM=$f timeout 300 python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras --dtype bf16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 fused_adam=$f', d['value'], d['ms_per_step'], d['config']['host_ms_per_step'], d.get('final_loss'))"
done
