#!/bin/bash
# rocprofv3 kernel trace of the default bench workload: per-kernel time per training step -> gpurun_out/profiles/<tag>_*
tag=${1:-r02}; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/profiles; mkdir -p $out
STEPS=10; WARM=3
rm -rf /tmp/prof_kt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o r -- python bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-eval "$@" > /tmp/prof_kt.log 2>&1
find /tmp/prof_kt -name "*kernel_stats.csv" -exec cp {} $out/${tag}_kernel_stats.csv \;
# steps of the traced command: warm-up, the timed region, (graph mode) the launch-by-launch region of the same length, the
# 5-step MFMA pass, two repeats of the timed region
python - $out/${tag}_kernel_stats.csv $((4*STEPS+WARM+5)) "$*" > $out/${tag}_kernel_summary.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); n = float(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-eval {sys.argv[3]}  ({int(n)} steps: warm-up, timed, launch by launch, MFMA pass, 2 repeats)")
print(f"total kernel time per training step: {tot/n/1e6:.3f} ms")
for r in rows[:45]:
    print(f"{float(r['TotalDurationNs'])/n/1e6:8.3f} ms/step {int(r['Calls'])/n:6.1f} calls/step  avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:110]}")
PY
tail -3 /tmp/prof_kt.log | cut -c1-400
