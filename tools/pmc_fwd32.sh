#!/bin/bash
# SQ / TA / TCP counter passes (rocprofv3 --kernel-trace --pmc, one pass per counter group) over tools/pmc_fwd32.py
# usage: tools/pmc_fwd32.sh <outfile> <knob 11 value> [key cin cout]
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$(realpath -m "$1"); shift
V=${1:-0}; shift
export TMPDIR=/tmp
cd /tmp
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
G2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
G3="SQ_WAIT_INST_LDS SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_WAVES SQ_ACTIVE_INST_MISC"
G4="TA_TA_BUSY TA_BUFFER_TOTAL_CYCLES"
G5="TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES"
G6="TCP_PENDING_STALL_CYCLES TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TCP_TA_DATA_STALL_CYCLES"
G7="GRBM_GUI_ACTIVE GRBM_COUNT"
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
i=0
for G in "$G1" "$G2" "$G3" "$G4" "$G5" "$G6" "$G7"; do
  i=$((i+1)); D=/tmp/d3d_pmc32_$i; rm -rf $D
  timeout 300 rocprofv3 --kernel-trace --pmc $G -d $D -- python $REPO/tools/pmc_fwd32.py $V 12 "$@" > $D.out 2> $D.err || { echo "pass $i ($G) failed: $(tail -2 $D.err | tr '\n' ' ')" >> "$OUT"; continue; }
  DB=$(find $D -name "*.db" | head -1)
  python - "$DB" >> "$OUT" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, counter_name, count(*), avg(counter_value), avg(duration)/1e3 from pmc_events where name like '%spconv_%' group by name, counter_name").fetchall()
for n, c, k, v, us in rows:
    print("%-34s %16.0f per launch  (%d launches, %.1f us)  %s" % (c, v, k, us, n[:90]))
PY
done
cat "$OUT"
