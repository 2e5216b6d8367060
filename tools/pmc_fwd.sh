#!/bin/bash
# SQ / TA / TCP / TCC counter passes over tools/pmc_fwd.py (the 64->64 forward kernel alone); one rocprofv3 --pmc pass
# per group, --kernel-trace only.  usage: tools/pmc_fwd.sh <out.txt>
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$(realpath -m "${1:-gpurun_out/pmc_fwd.txt}"); mkdir -p "$(dirname "$OUT")"
export TMPDIR=/tmp
cd /tmp
: > "$OUT"
GROUPS_=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"
 "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"
 "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU"
 "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE"
 "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD"
 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"
 "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"
 "TA_BUSY_avr" "TA_TA_BUSY_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum"
 "TCP_PENDING_STALL_CYCLES_sum" "TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_TA_TCP_STATE_READ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum"
 "TCP_TCC_READ_REQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum" "TCP_TOTAL_ACCESSES_sum" "TCP_GATE_EN1_sum" "TCP_GATE_EN2_sum"
 "TCC_REQ_sum" "TCC_HIT_sum" "TCC_MISS_sum" "TCC_TAG_STALL_sum" "TCC_EA0_RDREQ_sum" "TCC_BUSY_sum"
 "FETCH_SIZE" "GRBM_GUI_ACTIVE"
)
i=0
for G in "${GROUPS_[@]}"; do
  i=$((i+1)); D=/tmp/d3d_pmcf_$i; rm -rf $D
  if timeout 300 rocprofv3 --kernel-trace --pmc $G -d $D -- python $REPO/tools/pmc_fwd.py 40 > $D.out 2> $D.err; then
    DB=$(find $D -name "*.db" | head -1)
    python - "$DB" "$G" >> "$OUT" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
try:
    rows = cur.execute("select counter_name, count(*), sum(counter_value), avg(duration)/1e3, count(distinct dispatch_id) from pmc_events "
                       "where name like '%spconv_fwd_mfma_kernel%' group by counter_name").fetchall()
except sqlite3.OperationalError:
    rows = [r + (43,) for r in cur.execute("select counter_name, count(*), sum(counter_value), avg(duration)/1e3 from pmc_events "
                                           "where name like '%spconv_fwd_mfma_kernel%' group by counter_name").fetchall()]
for name, nrow, tot, dur, nd in rows:
    print("%-36s per-launch %16.1f   (rows %d, launches %d, avg kernel %.2f us under the counters)" % (name, tot / max(nd, 1), nrow, nd, dur))
PY
  else
    echo "pass [$G] failed: $(grep -i -m1 "error\|invalid\|not" $D.err | cut -c1-160)" >> "$OUT"
  fi
done
cat "$OUT"
