#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3j29
mkdir -p $O
bash tools/profile_step.sh $O/kernel_stats_c3.txt --workload c3 --burn-in 0 > $O/prof_c3.log 2>&1
head -40 $O/kernel_stats_c3.txt | cut -c1-170
