#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2j26; mkdir -p $O
export TMPDIR=/tmp
timeout 900 bash tools/profile_step.sh $O/kernel_stats.txt; grep -n "GPU busy\|bench line\|rb_\|d3d_scan" $O/kernel_stats.txt | cut -c1-160 | head -24
