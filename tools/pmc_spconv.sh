#!/bin/bash
# HBM traffic and MFMA-busy counters of the default bench step (GPU box).  Separate rocprofv3 --pmc passes, one
# counter each, with --kernel-trace only (the pool refuses --pmc together with other trace domains):
#   FETCH_SIZE / WRITE_SIZE   -> profiles/round6_pmc_{fetch,write}_size.txt + profiles/round6_pmc_spconv.json
#   SQ_VALU_MFMA_BUSY_CYCLES  -> profiles/round6_pmc_sq_valu_mfma_busy_cycles.txt (dense conv kernels: MFMA utilisation)
# usage: tools/pmc_spconv.sh <outdir>     (copy the results into profiles/ afterwards)
set -o pipefail
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$(realpath -m "${1:-gpurun_out/pmc}"); mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
CMD="python $REPO/bench.py --steps 3 --warmup 3 --burn-in 0 --no-cpu-baseline --no-launch-log --no-extras --windows 1"  # fixed step count: the passes must see the same dispatches
for C in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  rm -rf /tmp/d3d_pmc_$C
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/d3d_pmc_$C -- $CMD > /tmp/d3d_pmc_$C.out 2> /tmp/d3d_pmc_$C.err || { echo "pass $C failed"; tail -5 /tmp/d3d_pmc_$C.err; continue; }
  DB=$(find /tmp/d3d_pmc_$C -name "*.db" | head -1)
  python $REPO/tools/prof_summary.py pmc "$DB" $C "$OUT/round6_pmc_$(echo $C | tr 'A-Z' 'a-z').txt"
done
python $REPO/tools/pmc_json.py "$OUT" && head -40 "$OUT/round6_pmc_spconv.json"
