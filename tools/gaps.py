#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV (launch gaps), per training step."""
import csv, glob, sys
path, steps = sys.argv[1], float(sys.argv[2])
rows = []
for f in glob.glob(path + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
gaps = []
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if 0 < g < 200000:  # ignore the pauses between phases of the script
        gaps.append(g)
print(f"kernels {len(rows) / steps:.1f}/step  busy {busy / steps / 1e3:.1f} us/step  gaps {sum(gaps) / steps / 1e3:.1f} us/step "
      f"({len(gaps) / steps:.1f} gaps/step, median {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us)")
