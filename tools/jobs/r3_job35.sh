#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3j35; mkdir -p $O
bash tools/profile_step.sh $O/kernel_stats_c4.txt --workload c4 --burn-in 0 > $O/prof_c4.log 2>&1
head -12 $O/kernel_stats_c4.txt | cut -c1-150
