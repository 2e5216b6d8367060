#!/bin/bash
# MIOpen's find pass picks its solvers from one timing each: runs land anywhere between 558 and 745 frames/s.
# Record the user find-db of several find-mode runs, keep the best, and check that immediate mode with that db
# reproduces it (tools/jobs/gpu_job20.sh -> det3d_amd/miopen_db/).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2j20; mkdir -p $O
export TMPDIR=/tmp
best=0; bestdb=""
for i in 1 2 3 4 5 6; do
  export MIOPEN_USER_DB_PATH=$O/db_$i; mkdir -p $MIOPEN_USER_DB_PATH
  D3D_MIOPEN_FIND=1 timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras --no-launch-log > $O/find_$i.json 2> /dev/null
  v=$(python -c "import json;print(json.loads(open('$O/find_$i.json').read().strip().splitlines()[-1])['value'])")
  echo "find run $i: $v frames/s; db files: $(ls $MIOPEN_USER_DB_PATH | tr '\n' ' ')"
  if python -c "import sys; sys.exit(0 if $v > $best else 1)"; then best=$v; bestdb=$MIOPEN_USER_DB_PATH; fi
done
echo "best $best in $bestdb"
rm -rf $O/db_best; cp -r $bestdb $O/db_best
for j in 1 2 3; do
  rm -rf /tmp/dbv_$j; cp -r $O/db_best /tmp/dbv_$j
  MIOPEN_USER_DB_PATH=/tmp/dbv_$j D3D_MIOPEN_FIND=0 timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras --no-launch-log > $O/imm_$j.json 2> /dev/null
  echo "immediate mode + best db, run $j: $(python tools/show_bench.py $O/imm_$j.json | head -1)"
done
du -sh $O/db_best; ls -la $O/db_best
