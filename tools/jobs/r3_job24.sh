#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 300 python tools/diag_plan_time.py 2>&1 | grep DGRAD
