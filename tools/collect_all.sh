#!/bin/bash
# One GPU call for profiles/: bash tools/collect_all.sh r04  (from the repository root on a GPU box).
#   profiles/collect.sh            default bench line (with cpu_baseline: BASELINE.md 3's 10 + 30 protocol), kernel trace,
#                                  FETCH / WRITE_SIZE passes (the PMC file records the library hash it was collected on)
#   bench variants: 16 384 rays, bf16 MLP operands, both (BASELINE config 5), the occupancy-grid workload (config 4) as a
#   replayed hipGraph and launch by launch, the 'trained-like' table of SURVEY 8(d), the data-parallel backward priced on
#   one GPU (--dp-windows), and the rocprofv3 kernel trace of the config-4 step.
#   (profiles/collect_counters.sh — grid kernel counter groups — and profiles/collect_mfma.sh — MLP kernel counters — are
#    separate calls: WITH_COUNTERS=1 adds them.)
# Copy what you want judged from gpurun_out/profiles/ to profiles/.
tag=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/profiles; mkdir -p $out
timeout 1500 bash profiles/collect.sh $tag > gpurun_out/collect.log 2>&1
if [ -n "$WITH_COUNTERS" ]; then
  timeout 400 bash profiles/collect_counters.sh $tag > gpurun_out/cc.log 2>&1
  timeout 250 bash profiles/collect_mfma.sh $tag > gpurun_out/cm.log 2>&1
fi
b() { name=$1; shift; timeout 300 python bench.py "$@" > $out/${tag}_bench_$name.json 2> /dev/null; }
b 16384 --rays 16384 --no-cpu-baseline
b bf16 --mlp-dtype bf16 --no-cpu-baseline
b 16384_bf16 --rays 16384 --mlp-dtype bf16 --no-cpu-baseline
b nerfmvl --workload nerfmvl --steps 128 --warmup 16
b nerfmvl_nograph --workload nerfmvl --steps 128 --warmup 16 --no-graph
b trained --table trained --no-cpu-baseline
b dpwindows --dp-windows --no-cpu-baseline --no-eval
# kernel trace of the config-4 step (replayed graph): per-kernel time per step
rm -rf /tmp/prof_mvl
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mvl -o r -- python bench.py --workload nerfmvl --steps 128 --warmup 16 > /dev/null 2>&1
find /tmp/prof_mvl -name "*kernel_stats.csv" -exec cp {} $out/${tag}_nerfmvl_kernel_stats.csv \;
python - $out/${tag}_nerfmvl_kernel_stats.csv $((320+16+3*128+5)) > $out/${tag}_nerfmvl_kernel_summary.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); n = float(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"rocprofv3 --kernel-trace --stats -- python bench.py --workload nerfmvl --steps 128 --warmup 16   ({int(n)} steps: 320 settling + 16 warm-up + 3 x 128 timed + 5 launch-by-launch; the first 16 grid updates are full 128^3 sweeps)")
print(f"total kernel time per training step (all of the above averaged): {tot/n/1e6:.3f} ms")
for r in rows[:45]:
    print(f"{float(r['TotalDurationNs'])/n/1e3:8.1f} us/step {int(r['Calls'])/n:6.2f} calls/step  avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:110]}")
PY
ls -la $out
