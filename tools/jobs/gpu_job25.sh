#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2j25; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_spconv_gpu.py -q -m gpu -k "rulebook or plan or dense or strided or conv_rule" --tb=short 2>&1 | tail -3
timeout 900 bash tools/profile_step.sh $O/kernel_stats.txt; grep -n "GPU busy\|bench line\|rb_scan\|rb_conv" $O/kernel_stats.txt | cut -c1-140 | head -10
