#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3j25
mkdir -p $O
timeout 900 python -m pytest tests/test_optim_gpu.py tests/test_dist_gpu.py tests/test_e2e_gpu.py tests/test_graph_tail_gpu.py tests/test_determinism_gpu.py -q -m gpu --tb=short -x 2>&1 | tail -15
for f in 1 0 1 0; do
D3D_FUSED_ADAM=$f timeout 300 python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32 fused_adam=$f', d['value'], d['ms_per_step'], d['config']['host_ms_per_step'])"
done
for f in 1 0; do
D3D_FUSED_ADAM=$f timeout 300 python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras --dtype bf16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 fused_adam=$f', d['value'], d['ms_per_step'], d['config']['host_ms_per_step'])"
done
timeout 300 python tools/host_profile.py . fp32 > $O/host_profile_fp32.txt 2>&1; head -60 $O/host_profile_fp32.txt | cut -c1-150
