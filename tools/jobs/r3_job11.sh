#!/bin/bash
# round 3, job 11: deterministic loss scalars, dgrad BN-backward epilogue; graph-tail file; fp32 profile
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3j11
mkdir -p $O
timeout 900 python -m pytest tests/test_conv2d_gpu.py tests/test_loss_gpu.py tests/test_graph_tail_gpu.py tests/test_e2e_gpu.py -q -m gpu --tb=short 2>&1 | tail -30 > $O/tests.log
tail -8 $O/tests.log
bash tools/profile_step.sh $O/kernel_stats_fp32.txt --dtype fp32 > $O/prof_fp32.log 2>&1
head -42 $O/kernel_stats_fp32.txt | cut -c1-150
