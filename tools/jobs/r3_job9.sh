#!/bin/bash
# round 3, job 9: c4 loss at batch 16 (1960?) -- bisect batch size / precision; RCCL single-rank smoke; rpn probe
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3j9
mkdir -p $O
for cfg in "4 fp32" "16 fp32" "16 bf16" "8 fp32"; do timeout 300 python tools/diag_workload.py c4 $cfg 6 2>&1 | grep -E "step|Error|error" | tail -7; done > $O/diag_c4.txt
cat $O/diag_c4.txt
timeout 600 python -m pytest tests/test_dist_gpu.py -q -m gpu --tb=short 2>&1 | tail -15 > $O/tests.log
tail -6 $O/tests.log
timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --workload c2-60k 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(d.get('rpn_mfma'))[:1500])"
