#!/bin/bash
# Decisive form of tools/loop_rccl_bench.sh: the capture of the DP step is stalled for 300 ms (LNH_DEBUG_CAPTURE_STALL_MS), so
# a sweep of ProcessGroupNCCL's watchdog thread (every 100 ms) certainly falls into it.  Without the drain before the capture
# (LNH_CAPTURE_DRAIN_MS=0) the watchdog still holds the eager steps' collectives and polls their events while RCCL's stream
# is capturing: hipErrorCapturedEvent, abort.  With the drain (default 250 ms) it has nothing to poll.
N=${1:-6}; OUT=${2:-gpurun_out/r6/rccl_race}; mkdir -p "$OUT"
export LNH_DIST_BACKEND=nccl LNH_DP_SINGLE_RANK=1 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
for drain in 0 250; do
  for i in $(seq 1 "$N"); do
    export MASTER_PORT=$((20000 + RANDOM % 20000))
    LNH_DEBUG_CAPTURE_STALL_MS=300 LNH_CAPTURE_DRAIN_MS=$drain timeout 300 python bench.py --gpus 1 --steps 5 --warmup 2 \
        --no-eval --no-cpu-baseline --no-mfma-states --rays 1024 > "$OUT/d${drain}_$i.out" 2> "$OUT/d${drain}_$i.err"
    rc=$?
    why=$(grep -m1 -o "hipErrorCapturedEvent\|operation not permitted on an event last recorded in a capturing stream" "$OUT/d${drain}_$i.err" | head -1)
    g=$(python3 -c "
import json,sys
l=[x for x in open('$OUT/d${drain}_$i.out') if x.startswith('{')]
d=json.loads(l[0]) if l else {}
print('graph_error=%s' % d.get('graph',{}).get('error') if d else 'no line')")
    echo "stall=300ms drain=${drain}ms run $i rc=$rc $g $why" | tee -a "$OUT/summary.txt"
    rm -f "$OUT/d${drain}_$i.out"; [ "$rc" = 0 ] && rm -f "$OUT/d${drain}_$i.err" || { tail -c 3000 "$OUT/d${drain}_$i.err" > "$OUT/d${drain}_$i.tail"; rm -f "$OUT/d${drain}_$i.err"; }
  done
done
