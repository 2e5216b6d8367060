#!/bin/bash
# round 3, job 4: kernel traces of the bf16 and the fp32 (split RPN) step; fixed tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3j4
mkdir -p $O
timeout 300 python -m pytest tests/test_conv2d_gpu.py tests/test_determinism_gpu.py -q -m gpu --tb=short 2>&1 | tail -12 > $O/tests.log
tail -4 $O/tests.log
bash tools/profile_step.sh $O/kernel_stats_bf16.txt > $O/prof_bf16.log 2>&1
head -45 $O/kernel_stats_bf16.txt | cut -c1-150
bash tools/profile_step.sh $O/kernel_stats_fp32.txt --dtype fp32 > $O/prof_fp32.log 2>&1
head -45 $O/kernel_stats_fp32.txt | cut -c1-150
