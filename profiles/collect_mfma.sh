#!/bin/bash
# MFMA / issue counters of the four MLP kernels of a training step:  bash profiles/collect_mfma.sh r02
# (rocprofv3 --pmc passes over `python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-eval`, kernel trace only)
tag=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/profiles; mkdir -p $out
groups=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_F16"
 "GRBM_GUI_ACTIVE"
)
i=0
for g in "${groups[@]}"; do
  rm -rf /tmp/mf_$i
  timeout 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d /tmp/mf_$i -o r -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-eval > /tmp/mf_$i.log 2>&1 || echo "group $i failed"
  i=$((i+1))
done
python - $out/${tag}_mfma_counters.json <<'PY'
import csv, glob, json, sys, collections, re
res = collections.defaultdict(dict)
names = ("k_mlp_forward", "k_mlp_backward_wi", "k_color_forward", "k_color_backward_wi")
for d in sorted(glob.glob("/tmp/mf_*")):
    if d.endswith(".log"): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(float); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = next((n for n in names if n in r["Kernel_Name"]), None)
            if not k: continue
            agg[(k, r["Counter_Name"])] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for (k, c), v in agg.items():
            res[k][c] = round(v / cnt[(k, c)], 1)
for k, c in res.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        # MFMA-busy cycles are summed over the 1024 SIMDs of the chip; GUI_ACTIVE over the 8 XCDs
        c["mfma_busy_fraction"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (c["GRBM_GUI_ACTIVE"] / 8), 4)
json.dump({"command": "rocprofv3 --pmc <group> --kernel-trace -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-eval "
                      "(average per launch; k_mlp_forward runs twice per step: 3.15 M coarse + 0.26 M fine points)",
           "kernels": res}, open(sys.argv[1], "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
PY
