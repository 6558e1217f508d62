#!/bin/bash
# A/B of library variants built with LNH_VARIANT=<name> (lidar-nerf_amd/build.py): bash tools/ab.sh "<bench_grid args>" v1 v2 ...
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
args="$1"; shift
export LNH_ALLOW_VARIANT=1
for v in "$@"; do
  rm -rf /tmp/kt_$v
  LNH_LIB_PATH=$PWD/lidar-nerf_amd/lib/liblidarnerf_hip_$v.so rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$v -o r -- python tools/bench_grid.py --reps 5 $args > /tmp/ab_$v.log 2>&1
  echo "== $v"; python tools/ktrace.py /tmp/kt_$v 2>&1 | head -3 | grep k_grid; grep -i "error\|Traceback" /tmp/ab_$v.log | head -3
done > gpurun_out/ab.log 2>&1
