#!/bin/bash
# Regenerates the files of this directory on a GPU box:  bash profiles/collect.sh r01
#   <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-eval`
#   <tag>_kernel_summary.txt the same, per training step, top kernels
#   <tag>_pmc.json           FETCH_SIZE / WRITE_SIZE per launch of the grid kernels (separate --pmc passes, no other
#                            trace domains; FETCH_SIZE doubled per MI355X_MICROARCH.md "HBM": gfx950 tallies 128-B reads at 64 B)
#   <tag>_bench.json         the default bench line (with cpu_baseline)
tag=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/profiles; mkdir -p $out
STEPS=10; WARM=3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o r -- python bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-eval --no-mfma-states > /dev/null 2>&1
find /tmp/prof_kt -name "*kernel_stats.csv" -exec cp {} $out/${tag}_kernel_stats.csv \;
# (+5 steps: the MFMA pass bench.py runs after the timed region; + STEPS: the launch-by-launch region behind the replayed one;
#  + 2 x STEPS: the two repeats of the timed region it lists as spread; + 100: the 100-step region behind `value_100`)
python - $out/${tag}_kernel_stats.csv $((4*STEPS+WARM+5+100)) > $out/${tag}_kernel_summary.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); n = float(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-eval   ({int(n)} steps: warm-up, 10 replayed, 10 launch by launch, the 5-step MFMA pass, 2 x 10 repeats, the 100-step region)")
print(f"total kernel time per training step: {tot/n/1e6:.3f} ms")
for r in rows[:40]:
    print(f"{float(r['TotalDurationNs'])/n/1e6:8.3f} ms/step {int(r['Calls'])/n:6.1f} calls/step  avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:110]}")
PY
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/prof_$c -o r -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-eval --no-mfma-states --no-graph > /dev/null 2>&1
done
python - $out/${tag}_pmc.json <<'PY'
import csv, glob, json, sys, collections
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/prof_{c}/**/*counter_collection.csv", recursive=True)
    if not f: continue
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"]
        agg[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k in agg:
        short = next((s for s in ("k_grid_bwd_scatter_plain", "k_grid_bwd_scatter", "k_grid_bwd_reduce", "k_grid_forward", "k_color_backward_wi", "k_color_forward",
                                  "k_mlp_backward_wi", "k_mlp_forward") if s in k), None)
        if short: res[short][c + "_KB_per_launch_raw"] = round(agg[k] / cnt[k], 1)
for k, d in res.items():
    f, w = d.get("FETCH_SIZE_KB_per_launch_raw"), d.get("WRITE_SIZE_KB_per_launch_raw")
    if f is not None and w is not None:
        # gfx950 tallies 16-B/lane streaming reads at half (guide, "HBM"): only the reduce pass issues those.  The
        # 4-12 B/lane loads of the other kernels are calibrated on k_grid_bwd_scatter (raw FETCH_SIZE == known input bytes).
        d["fetch_scale"] = 2.0 if k == "k_grid_bwd_reduce" else 1.0
        d["hbm_bytes_per_launch"] = int((d["fetch_scale"] * f + w) * 1024)
import hashlib, os
lib = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "lidar-nerf_amd", "lib", "liblidarnerf_hip.so")
json.dump({"command": "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-eval --no-mfma-states --no-graph",
           "lib_sha16": hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16],   # bench.py compares it with the library it runs
           "note": "KB per launch, averaged over all launches. hbm_bytes_per_launch = fetch_scale*FETCH_SIZE + WRITE_SIZE "
                   "(fetch_scale 2 for the 16 B/lane streaming loads of k_grid_bwd_reduce, 1 elsewhere; see profiles/README.md)",
           "kernels": res},
          open(sys.argv[1], "w"), indent=1)
PY
timeout 900 python bench.py > $out/${tag}_bench.json 2> /dev/null
tail -c 600 $out/${tag}_bench.json; echo; cat $out/${tag}_pmc.json | head -40
