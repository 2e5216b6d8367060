#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_spconv_gpu.py tests/test_conv2d_gpu.py -q -m gpu -s 2>&1 | grep -E "worst|max \||lowest|rel L2|relative RMS|box_preds|passed|failed" | cut -c1-300
