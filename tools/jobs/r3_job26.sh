#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 300 python tools/diag_adam.py fp32 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/diag_adam.py bf16 2>&1 | grep -v amdgpu.ids
