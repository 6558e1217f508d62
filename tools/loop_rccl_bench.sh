#!/bin/bash
# The captured-DP bench line of tests/test_rccl_gpu.py::test_rccl_bench_single_rank_graph_line, N times in a row, each in a
# fresh process on a fresh rendezvous port: how often does it fail, and with what?  A failing run is repeated once with
# AMD_LOG_LEVEL=3 and its tail kept.   usage: tools/loop_rccl_bench.sh [N=50] [outdir=gpurun_out/r6/rccl_loop]
N=${1:-50}; OUT=${2:-gpurun_out/r6/rccl_loop}; mkdir -p "$OUT"
export LNH_DIST_BACKEND=nccl LNH_DP_SINGLE_RANK=1 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
fail=0
for i in $(seq 1 "$N"); do
  export MASTER_PORT=$((20000 + RANDOM % 20000))
  t0=$(date +%s.%N)
  timeout 300 python bench.py --gpus 1 --steps 5 --warmup 2 --no-eval --no-cpu-baseline --no-mfma-states --rays 1024 \
      > "$OUT/run_$i.out" 2> "$OUT/run_$i.err"
  rc=$?
  verdict=$(python3 - "$OUT/run_$i.out" <<'PY'
import json, sys
lines = [l for l in open(sys.argv[1]) if l.startswith("{")]
d = json.loads(lines[0]) if lines else {}
g = d.get("graph", {})
ok = d.get("n_gpus") == 1 and "graph" in d and "error" not in g and "collectives" in d.get("config", {}).get("launch", "")
print("ok" if ok else "BAD graph=%s launch=%s" % (g, d.get("config", {}).get("launch")))
PY
)
  echo "run $i rc=$rc $(python3 -c "print(round($(date +%s.%N)-$t0,1))")s $verdict" | tee -a "$OUT/summary.txt"
  if [ "$rc" != 0 ] || [ "$verdict" != ok ]; then
    fail=$((fail+1))
    tail -40 "$OUT/run_$i.err" > "$OUT/fail_$i.tail"
    AMD_LOG_LEVEL=3 timeout 300 python bench.py --gpus 1 --steps 5 --warmup 2 --no-eval --no-cpu-baseline --no-mfma-states \
        --rays 1024 > "$OUT/rerun_$i.out" 2> "$OUT/rerun_$i.err"; echo "  rerun rc=$?" | tee -a "$OUT/summary.txt"
    tail -c 20000 "$OUT/rerun_$i.err" > "$OUT/rerun_$i.tail"; rm -f "$OUT/rerun_$i.err"
  else
    rm -f "$OUT/run_$i.out" "$OUT/run_$i.err"
  fi
done
echo "failures: $fail of $N" | tee -a "$OUT/summary.txt"
