#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3j18
mkdir -p $O
timeout 300 python tools/tune_fwd_f32.py > $O/tune_fwd_f32.txt 2>&1
cat $O/tune_fwd_f32.txt | grep -v Warning
timeout 300 python -m pytest tests/test_spconv_gpu.py tests/test_conv2d_gpu.py -q -m gpu --tb=short -x 2>&1 | tail -3
timeout 300 python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-launch-log 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32', d['value'], d['ms_per_step'], d['config']['host_ms_per_step'])"
timeout 300 python bench.py --gpus 1 --steps 40 --warmup 5 --dtype bf16 --no-cpu-baseline --no-extras --no-launch-log 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16', d['value'], d['ms_per_step'], d['config']['host_ms_per_step'])"
