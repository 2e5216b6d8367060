#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3j17
mkdir -p $O
VARIANTS=8,9,0 WGRAD_WAVES=8,4 timeout 300 python tools/bench_conv2d.py --shape 4 200 176 128 128 3 1 1 > $O/bench_conv2d.txt 2>&1
grep shape $O/bench_conv2d.txt
timeout 300 python -m pytest tests/test_conv2d_gpu.py -q -m gpu --tb=short -x 2>&1 | tail -3
D3D_CONV2D_WGRAD_WAVES=4 D3D_CONV2D_VARIANT=0 timeout 300 python -m pytest tests/test_conv2d_gpu.py -q -m gpu --tb=short -x 2>&1 | tail -2
timeout 300 python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-launch-log 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32', d['value'], d['ms_per_step'], d['config']['host_ms_per_step'])"
timeout 300 python bench.py --gpus 1 --steps 40 --warmup 5 --dtype bf16 --no-cpu-baseline --no-extras --no-launch-log 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16', d['value'], d['ms_per_step'], d['config']['host_ms_per_step'])"
