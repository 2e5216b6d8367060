#!/bin/bash
# The one gpurun entry point of this repository:  gpurun -- 'bash tools/gpu_job.sh <tag> <stage> [<stage> ...]'
# Every stage writes under gpurun_out/<tag>/ (merged back by gpurun); what is worth keeping is copied into profiles/
# by hand afterwards.  Stages:
#   tests            the driver's round-end sequence: pytest -m gpu -x, smoke()
#   tests:<expr>     pytest -m gpu -k <expr>
#   env:<VAR=VAL[,..]> one 60-step bench.py run under these environment variables
#   bench            bench.py at the driver's flags (--gpus 1 --steps 20 --warmup 5)
#   bench300         300 timed steps, no extras
#   multirank        VERDICT r3 item 1: torchrun x2 over gloo on the one GPU (c2 / c3 / c4), 1-rank RCCL bench, and the
#                    per-rank host budget (taskset to cores/8)
#   prof[:<workload>]  rocprofv3 --kernel-trace --stats of the bench step -> kernel + gap tables
#   pmc              separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES) -> <tag>/pmc/round4_pmc_*
#   wl:<workload>    bench.py --workload c4|c5|c3|c2-60k, no extras
#   conv2d           tools/bench_conv2d.py (dense conv kernel variants)
#   wgradstream      the weight-gradient stream off / on, plain and under a 1-rank RCCL communicator
#   py:<script> ...  python <script> (rest of the line), log to <tag>/<script>.log
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-job}; shift
O=$PWD/gpurun_out/$TAG
mkdir -p "$O"
port() { python -c "import socket; s=socket.socket(); s.bind(('127.0.0.1',0)); print(s.getsockname()[1])"; }
show() { python tools/show_bench.py "$1" 2>/dev/null | head -${2:-3}; }

while [ $# -gt 0 ]; do
  st=$1; shift
  echo "=== stage $st"
  case "$st" in
    tests)
      timeout 1800 python -m pytest tests/ -x -q -m gpu -rP --durations=25 > $O/tests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/tests.log; grep -A 27 "slowest" $O/tests.log | cut -c1-150
      grep -h "vs float64\|max |pred - reference|\|worst gradient-norm\|ResNetFHD\|distance from float64\|forward split vs\|300 steps\|medians: split\|steady-state steps\|max err vs float64\|losses, tile kernels\|relative loss difference\|held-out raw\|reference detections matched\|fused SyncBN nodes\|frames at the max_voxels\|rulebooks, " $O/tests.log | grep -v "print(" | cut -c1-900 > $O/tests_figures.txt; cat $O/tests_figures.txt
      timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 ;;
    testsall:*)
      timeout 1500 python -m pytest tests/ -q -m gpu -rP -k "${st#testsall:}" > $O/tests_all.log 2>&1; echo "pytest rc=$?"
      grep -n "^FAILED\|^E  .*Error\|passed\|failed" $O/tests_all.log | cut -c1-260 | head -60
      grep -h "vs float64\|max |pred - reference|\|worst gradient-norm\|ResNetFHD\|distance from float64\|forward split vs\|300 steps\|medians: split\|steady-state steps\|max err vs float64\|losses, tile kernels\|relative loss difference\|held-out raw\|reference detections matched\|fused SyncBN nodes" $O/tests_all.log | grep -v "print(" | cut -c1-900 > $O/tests_all_figures.txt; cat $O/tests_all_figures.txt ;;
    tests:*)
      timeout 1500 python -m pytest tests/ -x -q -m gpu -k "${st#tests:}" > $O/tests_k.log 2>&1; echo "pytest rc=$?"; tail -15 $O/tests_k.log ;;
    bench)
      timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench.err; echo "bench rc=$?"
      show $O/bench_driver_flags.json 6 ;;
    bench300)
      timeout 400 python bench.py --gpus 1 --steps 300 --warmup 10 --no-cpu-baseline --no-extras > $O/bench_300steps.json 2> $O/bench300.err
      show $O/bench_300steps.json 2 ;;
    multirank)
      NC=$(nproc); PER=$(( NC / 8 )); [ $PER -lt 1 ] && PER=1
      echo "host cores: $NC, per-rank share on an 8-GPU node: $PER" | tee $O/multirank.txt
      B="--steps 20 --warmup 3 --no-cpu-baseline --no-extras"
      # (a) plain single process, the reference point
      timeout 300 python bench.py --gpus 1 $B > $O/mr_plain.json 2> $O/mr_plain.err
      # (b) the same with every collective of the step issued on a 1-rank RCCL communicator; bucket-count and
      #     record-stream variants (D3D_DP_BUCKETS=0: one flat all-reduce after backward)
      R1="D3D_DIST_SINGLE=1 D3D_DIST_BACKEND=nccl RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1"
      env $R1 MASTER_PORT=$(port) timeout 300 python bench.py --gpus 1 $B > $O/mr_rccl1.json 2> $O/mr_rccl1.err
      env $R1 MASTER_PORT=$(port) D3D_DP_BUCKETS=1 timeout 300 python bench.py --gpus 1 $B > $O/mr_rccl1_b1.json 2> $O/mr_rccl1_b1.err
      env $R1 MASTER_PORT=$(port) D3D_DP_BUCKETS=0 timeout 300 python bench.py --gpus 1 $B > $O/mr_rccl1_b0.json 2> $O/mr_rccl1_b0.err
      env $R1 MASTER_PORT=$(port) TORCH_NCCL_AVOID_RECORD_STREAMS=1 timeout 300 python bench.py --gpus 1 $B > $O/mr_rccl1_nors.json 2> $O/mr_rccl1_nors.err
      # where the 1-rank cost comes from: more hardware queues; the exchanging step without the collectives; the group alone
      env $R1 MASTER_PORT=$(port) GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --gpus 1 $B > $O/mr_rccl1_q8.json 2> $O/mr_rccl1_q8.err
      env $R1 MASTER_PORT=$(port) D3D_DIST_NOCOLL=1 timeout 300 python bench.py --gpus 1 $B > $O/mr_rccl1_nocoll.json 2> $O/mr_rccl1_nocoll.err
      env $R1 D3D_DIST_SINGLE=2 MASTER_PORT=$(port) timeout 300 python bench.py --gpus 1 $B > $O/mr_group_only.json 2> $O/mr_group_only.err
      GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --gpus 1 $B > $O/mr_plain_q8.json 2> $O/mr_plain_q8.err
      # (c) the per-rank host budget: the process and its threads pinned to cores/8 CPUs
      taskset -c 0-$((PER-1)) timeout 300 python bench.py --gpus 1 $B > $O/mr_taskset.json 2> $O/mr_taskset.err
      env $R1 MASTER_PORT=$(port) taskset -c 0-$((PER-1)) timeout 300 python bench.py --gpus 1 $B > $O/mr_taskset_rccl1.json 2> $O/mr_taskset_rccl1.err
      if [ "$MR_PROF" = "1" ]; then
        export D3D_PROFILE_TIMELINE=1
        env $R1 MASTER_PORT=$(port) bash tools/profile_step.sh $O/kernel_stats_rccl1.txt > $O/prof_rccl1.log 2>&1
        env $R1 MASTER_PORT=$(port) D3D_DIST_NOCOLL=1 bash tools/profile_step.sh $O/kernel_stats_nocoll.txt > $O/prof_nocoll.log 2>&1
        bash tools/profile_step.sh $O/kernel_stats_plain.txt > $O/prof_plain.log 2>&1
      fi
      # (d) two ranks under torch.distributed.run on ONE GPU over gloo: the driver's launch line
      for wl in c2-20k c4 c3; do
        D3D_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
          --master-port $(port) bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline --workload $wl \
          > $O/mr_torchrun2_$wl.json 2> $O/mr_torchrun2_$wl.err; echo "torchrun2 $wl rc=$?"
      done
      python - "$O" <<'PY' | tee -a $O/multirank.txt
import json, sys, os
O = sys.argv[1]
def line(f):
    try:
        return json.loads([l for l in open(os.path.join(O, f)) if l.startswith("{")][-1])
    except Exception as e:
        return None
for f in ("mr_plain", "mr_rccl1", "mr_rccl1_b1", "mr_rccl1_b0", "mr_rccl1_nors", "mr_rccl1_q8", "mr_rccl1_nocoll", "mr_group_only", "mr_plain_q8", "mr_taskset", "mr_taskset_rccl1", "mr_torchrun2_c2-20k", "mr_torchrun2_c4", "mr_torchrun2_c3"):
    d = line(f + ".json")
    if d is None:
        print("%-22s FAILED (see %s.err)" % (f, f)); continue
    c = d["config"]
    print("%-22s n_gpus=%d %8.1f frames/s %7.3f ms/step  host %s  loss %s -> %s" % (
        f, d["n_gpus"], d["value"], d["ms_per_step"], c.get("host_ms_per_step"), c.get("first_loss"), c.get("final_loss")))
PY
      ;;
    geoprof|geoprof:*)
      # the geometry stage alone under rocprofv3 (isolated kernel durations; in the step these kernels share the CUs with
      # the main stream's): chain entry points vs the per-layer ones
      wl=${st#geoprof}; wl=${wl#:}; wl=${wl:-c2-20k}
      export TMPDIR=/tmp
      for ch in 1 0; do
        rm -rf /tmp/d3d_geo && mkdir -p /tmp/d3d_geo
        R=$PWD
        (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/d3d_geo -- python $R/tools/geometry_bench.py $wl $ch 30 > $O/geo_${wl}_chain$ch.log 2>&1)
        DB=$(find /tmp/d3d_geo -name "*.db" | head -1)
        MS=$(grep "ms wall" $O/geo_${wl}_chain$ch.log | sed 's/.*: \([0-9.]*\) ms wall.*/\1/' | python -c "import sys; print(float(sys.stdin.read().strip() or 1) * 30)")
        python tools/prof_summary.py kernels "$DB" "$MS" 30 $O/geo_kernels_${wl}_chain$ch.txt "tools/geometry_bench.py $wl chain=$ch, 30 builds" "$(grep 'ms wall' $O/geo_${wl}_chain$ch.log)"
        head -24 $O/geo_kernels_${wl}_chain$ch.txt | cut -c1-150
      done ;;
    syncbn)
      # nuScenes CBGS with norm_cfg SyncBN through the fused nodes, every collective on a 1-rank RCCL
      # communicator, against the plain-BN step under the same communicator and the plain single-process step
      B="--steps 10 --warmup 3 --burn-in 0.5 --no-cpu-baseline --no-extras --workload c4"
      R1="D3D_DIST_SINGLE=1 D3D_DIST_BACKEND=nccl RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1"
      timeout 400 python bench.py --gpus 1 $B > $O/sb_plain.json 2> $O/sb_plain.err
      env $R1 MASTER_PORT=$(port) timeout 400 python bench.py --gpus 1 $B > $O/sb_rccl1_bn.json 2> $O/sb_rccl1_bn.err
      env $R1 MASTER_PORT=$(port) timeout 400 python bench.py --gpus 1 $B --syncbn > $O/sb_rccl1_syncbn.json 2> $O/sb_rccl1_syncbn.err
      env $R1 MASTER_PORT=$(port) D3D_FUSE_STACK=0 D3D_FUSE_BLOCK=0 D3D_DENSE_CONV=0 timeout 400 python bench.py --gpus 1 $B --syncbn > $O/sb_rccl1_syncbn_unfused.json 2> $O/sb_rccl1_syncbn_unfused.err
      python - "$O" <<'PY' | tee $O/syncbn.txt
import json, sys, os
O = sys.argv[1]
for f in ("sb_plain", "sb_rccl1_bn", "sb_rccl1_syncbn", "sb_rccl1_syncbn_unfused"):
    try:
        d = json.loads([l for l in open(os.path.join(O, f + ".json")) if l.startswith("{")][-1])
        c = d["config"]
        print("%-26s %8.1f frames/s %8.3f ms/step  host %s  loss %s -> %s  norm: %s" % (
            f, d["value"], d["ms_per_step"], c.get("host_ms_per_step"), c.get("first_loss"), c.get("final_loss"), c.get("norm")))
    except Exception as e:
        print("%-26s FAILED (%s)" % (f, e))
PY
      ;;
    wgradstream)
      # the weight-gradient stream (sparse_stack.WGRAD_STREAM) off / on, plain and next to a 1-rank RCCL process group
      B="--gpus 1 --steps 40 --warmup 3 --no-cpu-baseline --no-extras"
      R1="D3D_DIST_SINGLE=1 D3D_DIST_BACKEND=nccl RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1"
      for rep in 1 2; do for w in 0 1; do
        D3D_WGRAD_STREAM=$w timeout 300 python bench.py $B > $O/ws_plain_${w}_$rep.json 2> $O/ws_plain_${w}_$rep.err
        env $R1 MASTER_PORT=$(port) D3D_WGRAD_STREAM=$w timeout 300 python bench.py $B > $O/ws_rccl1_${w}_$rep.json 2> $O/ws_rccl1_${w}_$rep.err
        echo "wgrad stream $w: plain $(show $O/ws_plain_${w}_$rep.json 1 | cut -c1-45)  1-rank RCCL $(show $O/ws_rccl1_${w}_$rep.json 1 | cut -c1-45)"
      done; done | tee $O/wgradstream.txt ;;
    mrprio)
      # does the geometry stream keep its priority next to an RCCL process group?  (profiles/round4_multirank.txt)
      B="--steps 20 --warmup 3 --no-cpu-baseline --no-extras"
      R1="D3D_DIST_SINGLE=1 D3D_DIST_BACKEND=nccl RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1"
      run() { n=$1; shift; env "$@" MASTER_PORT=$(port) timeout 300 python bench.py --gpus 1 $B > $O/p_$n.json 2> $O/p_$n.err; }
      run plain X=1
      run plain_prio0 D3D_PREFETCH_PRIORITY=0
      run rccl1_reserved $R1
      run rccl1_late $R1 D3D_PREFETCH_RESERVE=0
      run rccl1_late_hp $R1 D3D_PREFETCH_RESERVE=0 TORCH_NCCL_HIGH_PRIORITY=1
      run nocoll_reserved $R1 D3D_DIST_NOCOLL=1
      B="$B --no-prefetch-thread"; run rccl1_reserved_nothread $R1
      python - "$O" <<'PY' | tee $O/mrprio.txt
import json, sys, os, glob
O = sys.argv[1]
for f in sorted(glob.glob(os.path.join(O, "p_*.json")), key=os.path.getmtime):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print("%-26s %8.1f frames/s %7.3f ms/step  host %s" % (os.path.basename(f)[2:-5], d["value"], d["ms_per_step"], d["config"].get("host_ms_per_step")))
    except Exception as e:
        print("%-26s FAILED" % os.path.basename(f))
PY
      ;;
    graphab)
      # hipGraph tail (neck + head + loss) against the eager tail, both precisions (profiles/round4_graph_ab.txt)
      B="--steps 40 --warmup 3 --no-cpu-baseline --no-extras"
      for dt in fp32 bf16; do for g in 0 1; do
        D3D_GRAPH_TAIL=$g timeout 300 python bench.py --gpus 1 $B --dtype $dt > $O/g_${dt}_$g.json 2> $O/g_${dt}_$g.err
      done; done
      python - "$O" <<'PY' | tee $O/graph_ab.txt
import json, sys, os
O = sys.argv[1]
for dt in ("fp32", "bf16"):
    for g in (0, 1):
        try:
            d = json.loads([l for l in open(os.path.join(O, "g_%s_%d.json" % (dt, g))) if l.startswith("{")][-1])
            print("%-5s graph_tail=%d (captured %s) %8.1f frames/s %7.3f ms/step  host %s" % (
                dt, g, d["config"]["graphed_tail"], d["value"], d["ms_per_step"], d["config"]["host_ms_per_step"]))
        except Exception as e:
            print(dt, g, "FAILED", e)
PY
      ;;
    alloc)
      # caching-allocator growth inside the timed window, with and without size rounding
      B="--steps 20 --warmup 3 --no-cpu-baseline --no-extras"
      for v in none "roundup_power2_divisions:4" "roundup_power2_divisions:8" "expandable_segments:True"; do
        n=$(echo $v | tr -c 'a-zA-Z0-9' '_')
        if [ "$v" = none ]; then timeout 300 python bench.py --gpus 1 $B > $O/a_$n.json 2> $O/a_$n.err
        else PYTORCH_HIP_ALLOC_CONF=$v PYTORCH_CUDA_ALLOC_CONF=$v timeout 300 python bench.py --gpus 1 $B > $O/a_$n.json 2> $O/a_$n.err; fi
        echo "$v: $(grep -h 'timed region done' $O/a_$n.err | cut -c1-220)"; show $O/a_$n.json 1
      done ;;
    abtree|abtree:*)
      # this tree against another checkout of the repository that travelled with the snapshot (default ab_base/: `git worktree
      # add ab_base <commit>` + its own build), alternating, same box:  abtree[:<dir>]
      d=${st#abtree}; d=${d#:}; d=${d:-ab_base}
      for rep in 1 2 3; do for t in base new; do
        if [ $t = base ]; then b=$d/bench.py; else b=bench.py; fi
        timeout 300 python $b --gpus 1 --steps 60 --warmup 3 --no-cpu-baseline --no-extras > $O/abtree_${t}_$rep.json 2> $O/abtree_${t}_$rep.err
        echo "$t: $(show $O/abtree_${t}_$rep.json 2 | tr '\n' ' ' | cut -c1-120)"
      done; done | tee $O/abtree.txt ;;
    ab:*)
      # bench.py with an environment switch off and on, twice each, same box:  ab:D3D_PACK_ALL
      v=${st#ab:}
      for rep in 1 2; do for val in 0 1; do
        env $v=$val timeout 300 python bench.py --gpus 1 --steps 60 --warmup 3 --no-cpu-baseline --no-extras > $O/ab_${v}_${val}_$rep.json 2> $O/ab_${v}_${val}_$rep.err
        echo "$v=$val: $(show $O/ab_${v}_${val}_$rep.json 2 | tr '\n' ' ' | cut -c1-200)"
      done; done ;;
    allocx)
      # allocator experiment: the driver's 20-step window under caching-allocator settings; the line of
      # interest is bench.py's "allocator during it" note
      i=0
      for conf in "" "roundup_power2_divisions:8" "expandable_segments:True" "max_split_size_mb:32" "roundup_power2_divisions:8,max_split_size_mb:32" "expandable_segments:True,roundup_power2_divisions:8"; do
        i=$((i+1))
        PYTORCH_HIP_ALLOC_CONF="$conf" PYTORCH_ALLOC_CONF="$conf" timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/allocx_$i.json 2> $O/allocx_$i.err
        echo "[$conf] $(show $O/allocx_$i.json 1 | cut -c1-120)"; grep "c2-20k fp32 timed region" $O/allocx_$i.err | cut -c1-260; grep -i "warn\|not supported" $O/allocx_$i.err | head -3
      done ;;
    env:*)
      # one 60-step bench.py run under VAR=VALUE (comma-separated list):  env:D3D_BN_STREAM_BLOCKS=1024
      v=${st#env:}
      env ${v//,/ } timeout 300 python bench.py --gpus 1 --steps 60 --warmup 3 --no-cpu-baseline --no-extras > $O/env_${v//[^A-Za-z0-9_=]/_}.json 2> $O/env_${v//[^A-Za-z0-9_=]/_}.err
      echo "$v: $(show $O/env_${v//[^A-Za-z0-9_=]/_}.json 2 | tr '\n' ' ' | cut -c1-200)" ;;
    mrprof)
      export D3D_PROFILE_TIMELINE=1
      R1="D3D_DIST_SINGLE=1 D3D_DIST_BACKEND=nccl RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1"
      env $R1 MASTER_PORT=$(port) D3D_DIST_NOCOLL=1 bash tools/profile_step.sh $O/kernel_stats_nocoll.txt > $O/prof_nocoll.log 2>&1
      bash tools/profile_step.sh $O/kernel_stats_plain.txt > $O/prof_plain.log 2>&1
      head -5 $O/kernel_stats_nocoll.txt | cut -c1-200 ;;
    pmc)
      bash tools/pmc_spconv.sh $O/pmc > $O/pmc.log 2>&1; tail -30 $O/pmc.log | cut -c1-200 ;;
    wl:*)
      w=${st#wl:}
      timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --burn-in 0.5 --workload $w --no-cpu-baseline --no-extras > $O/wl_$w.json 2> $O/wl_$w.err; echo "workload $w rc=$?"
      show $O/wl_$w.json 14 ;;
    prof|prof:*)
      wl=${st#prof}; wl=${wl#:}; wl=${wl:-c2-20k}
      bash tools/profile_step.sh $O/kernel_stats_$wl.txt --workload $wl; head -45 $O/kernel_stats_$wl.txt ;;
    conv2d)
      timeout 600 python tools/bench_conv2d.py > $O/conv2d.txt 2>&1; tail -40 $O/conv2d.txt | cut -c1-900 ;;
    py:*)
      s=${st#py:}; n=$(basename $s .py)
      timeout 900 python $s "$@" > $O/$n.log 2>&1; echo "$s rc=$?"; tail -40 $O/$n.log; set -- ;;
    *) echo "unknown stage $st" ;;
  esac
done
