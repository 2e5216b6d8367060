#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2j21; mkdir -p $O
export TMPDIR=/tmp
for j in 1 2 3; do
  rm -rf /tmp/dbv_$j; cp -r det3d_amd/miopen_db /tmp/dbv_$j
  MIOPEN_USER_DB_PATH=/tmp/dbv_$j D3D_MIOPEN_FIND=1 timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras --no-launch-log > $O/finddb_$j.json 2> /dev/null
  echo "find mode + recorded db, run $j: $(python tools/show_bench.py $O/finddb_$j.json | head -1)"
  cmp -s /tmp/dbv_$j/*.ufdb.txt det3d_amd/miopen_db/*.ufdb.txt && echo "   (find-db unchanged)" || echo "   (find-db was rewritten)"
done
