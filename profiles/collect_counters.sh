#!/bin/bash
# PMC evidence for the hash-grid kernels (forward / backward scatter / backward reduce) at the BASELINE shape:
#   bash profiles/collect_counters.sh r02      (on a GPU box; writes gpurun_out/profiles/<tag>_grid_counters.json)
# One rocprofv3 --pmc pass per counter group (no other trace domains), workload = tools/bench_grid.py (3.4 M points,
# fp16 tables, LiDAR ray geometry, whole forward + whole backward, 3 launches each).
tag=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/profiles; mkdir -p $out
groups=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
 "SQ_INSTS_LDS_ATOMIC SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
 "TCP_TOTAL_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
 "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "GRBM_GUI_ACTIVE TA_BUSY_avr TCC_BUSY_avr"
)
i=0
for g in "${groups[@]}"; do
  rm -rf /tmp/pmc_$i
  timeout 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d /tmp/pmc_$i -o r -- python tools/bench_grid.py --reps 2 > /tmp/pmc_$i.log 2>&1 || echo "group $i failed: $g"
  i=$((i+1))
done
python - $out/${tag}_grid_counters.json <<'PY'
import csv, glob, json, sys, collections, re
res = collections.defaultdict(dict)
for d in sorted(glob.glob("/tmp/pmc_*")):
    if d.endswith(".log"): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(float); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_grid_[a-z_]+)", r["Kernel_Name"])
            if not m: continue
            key = (m.group(1), r["Counter_Name"])
            agg[key] += float(r["Counter_Value"]); cnt[key] += 1
        for (k, c), v in agg.items():
            res[k][c] = round(v / cnt[(k, c)], 1)
json.dump({"command": "rocprofv3 --pmc <group> --kernel-trace -- python tools/bench_grid.py --reps 2  (one pass per group; "
                      "values = average per launch; whole forward / whole backward over 3 407 872 points, fp16 tables)",
           "kernels": res}, open(sys.argv[1], "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
PY
