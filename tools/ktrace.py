#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel name (shortened) calls / total / average / min, optionally the
individual launches in order (--seq) — used with tools/bench_grid.py --per-level to split the backward per level."""
import csv
import glob
import re
import sys


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name[:60]


def main():
    path = sys.argv[1]
    files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True) if not path.endswith(".csv") else [path]
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    agg = {}
    for r in rows:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        agg.setdefault(short(r["Kernel_Name"]), []).append(d)
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k:32s} calls {len(v):5d}  total {sum(v) / 1e3:9.3f} ms  avg {sum(v) / len(v):9.1f} us  min {min(v):9.1f} us")
    if "--seq" in sys.argv:
        pat = sys.argv[sys.argv.index("--seq") + 1]
        for r in rows:
            if pat in r["Kernel_Name"]:
                print(f"{short(r['Kernel_Name']):28s} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f} us  grid {r.get('Grid_Size_X', '?')}")


if __name__ == "__main__":
    main()
