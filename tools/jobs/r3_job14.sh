#!/bin/bash
# round 3, job 14: head convolutions on the library's kernels
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3j14
mkdir -p $O
timeout 900 python -m pytest tests/test_conv2d_gpu.py tests/test_determinism_gpu.py tests/test_e2e_gpu.py tests/test_graph_tail_gpu.py tests/test_nms_gpu.py -q -m gpu --tb=short 2>&1 | tail -30 > $O/tests.log
tail -8 $O/tests.log
timeout 300 python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-launch-log 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32', d['value'], d['ms_per_step'], d['config']['host_ms_per_step'])"
timeout 300 python bench.py --gpus 1 --steps 40 --warmup 5 --dtype bf16 --no-cpu-baseline --no-extras --no-launch-log 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16', d['value'], d['ms_per_step'], d['config']['host_ms_per_step'])"
bash tools/profile_step.sh $O/kernel_stats_fp32.txt --dtype fp32 > $O/prof_fp32.log 2>&1
grep -E "launches/step|head_conv|igemm|SubTensor" $O/kernel_stats_fp32.txt | cut -c1-130
