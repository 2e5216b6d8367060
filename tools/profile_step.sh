#!/bin/bash
# rocprofv3 kernel trace of the default bench step, summarised on the GPU box (the rocpd .db is too big to
# copy back).  usage: tools/profile_step.sh <out.txt> [extra bench args...]
set -o pipefail
REPO=${D3D_PROFILE_REPO:-$(cd "$(dirname "$0")/.." && pwd)}
TOOLS=$(cd "$(dirname "$0")" && pwd)
OUT=$(realpath -m "$1"); shift
export TMPDIR=/tmp
rm -rf /tmp/d3d_prof && mkdir -p /tmp/d3d_prof
cd /tmp
STEPS=10
CMD="python $REPO/bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --no-launch-log --no-extras --windows 1 $*"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/d3d_prof -- $CMD > /tmp/d3d_prof/bench.out 2> /tmp/d3d_prof/bench.err || { tail -20 /tmp/d3d_prof/bench.err; exit 1; }
LINE=$(grep '"metric"' /tmp/d3d_prof/bench.out | tail -1)
DB=$(find /tmp/d3d_prof -name "*.db" | head -1)
MS=$(python -c "import json,sys; print(json.loads(sys.argv[1])['ms_per_step']*$STEPS)" "$LINE")
python $TOOLS/prof_summary.py kernels "$DB" "$MS" $STEPS "$OUT" "rocprofv3 --kernel-trace --stats -- $CMD" "bench line under the profiler: $LINE"
python $TOOLS/prof_summary.py gaps "$DB" "$MS" $STEPS "$OUT"
WL=$(python -c "import json,sys; c=json.loads(sys.argv[1])['config']; print(c['name'])" "$LINE")
DT=$(python -c "import json,sys; print('bf16' if json.loads(sys.argv[1])['dtype']=='bf16' else 'fp32')" "$LINE")
python $TOOLS/prof_summary.py roofjson "$DB" "$MS" $STEPS "${OUT%.txt}_roofline.json" "$WL" "$DT"
if [ -n "$D3D_PROFILE_TIMELINE" ]; then
  # the last two steps, kernel by kernel
  python $TOOLS/prof_summary.py timeline "$DB" "$(python -c "print($MS / $STEPS * 2.0)")" "${OUT%.txt}_timeline.txt"
fi
