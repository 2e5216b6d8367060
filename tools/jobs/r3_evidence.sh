#!/bin/bash
# round 3 evidence: PMC passes of the roofline kernel, kernel traces of both precisions, the bench as the driver runs it
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3ev
mkdir -p $O
bash tools/pmc_spconv.sh $O > $O/pmc.log 2>&1
tail -3 $O/pmc.log
bash tools/profile_step.sh $O/round3_kernel_stats_fp32.txt --dtype fp32 > $O/prof_fp32.log 2>&1
bash tools/profile_step.sh $O/round3_kernel_stats_bf16.txt --dtype bf16 > $O/prof_bf16.log 2>&1
head -12 $O/round3_kernel_stats_fp32.txt | cut -c1-140
