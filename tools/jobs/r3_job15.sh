#!/bin/bash
# round 3, job 15: head wgrad with the separate reduce; driver-flag bench; kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3j15
mkdir -p $O
timeout 600 python -m pytest tests/test_conv2d_gpu.py tests/test_determinism_gpu.py -q -m gpu --tb=short 2>&1 | tail -5 > $O/tests.log
tail -3 $O/tests.log
bash tools/profile_step.sh $O/kernel_stats_fp32.txt --dtype fp32 > $O/prof_fp32.log 2>&1
grep -E "launches/step|head_conv" $O/kernel_stats_fp32.txt | cut -c1-130
T0=$(date +%s)
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench wall $(( $(date +%s) - T0 )) s"
python tools/show_bench.py $O/bench.json workloads | cut -c1-420
