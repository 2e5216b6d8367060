#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2j11; mkdir -p $O
export TMPDIR=/tmp
echo "== tune fwd"; timeout 600 python tools/tune_fwd.py $O/tune_fwd.txt 2>&1 | tail -30
