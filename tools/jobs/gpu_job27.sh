#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_spconv_gpu.py tests/test_e2e_gpu.py tests/test_voxelize_gpu.py -q -m gpu --tb=short -x 2>&1 | tail -4
