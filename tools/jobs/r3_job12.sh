#!/bin/bash
# round 3, job 12: asm transposing reads in the dense wgrad; graph-tail file with deterministic MIOpen heads
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3j12
mkdir -p $O
timeout 900 python -m pytest tests/test_conv2d_gpu.py tests/test_graph_tail_gpu.py -q -m gpu --tb=short 2>&1 | tail -30 > $O/tests.log
tail -8 $O/tests.log
timeout 200 python tools/bench_conv2d.py --shape 4 200 176 128 128 3 1 1 > $O/bench_conv2d.txt 2>&1
grep shape $O/bench_conv2d.txt
timeout 300 python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-launch-log 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32', d['value'], d['ms_per_step'], d['config']['host_ms_per_step'])"
timeout 300 python bench.py --gpus 1 --steps 40 --warmup 5 --dtype bf16 --no-cpu-baseline --no-extras --no-launch-log 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16', d['value'], d['ms_per_step'], d['config']['host_ms_per_step'])"
