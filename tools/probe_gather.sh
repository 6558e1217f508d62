#!/bin/bash
# Gather cost model of the encode forward.  One library per probe (tools/probe_variants.py gather: where the
# second load of every lane of a hashed level goes), the forward of tools/bench_grid.py timed under rocprofv3:
#   bash tools/probe_gather.sh build     where hipcc is (the variant libraries travel with the gpurun snapshot)
#   bash tools/probe_gather.sh run       on a GPU box, from the repository root -> gpurun_out/ab.log
#   bash tools/probe_gather.sh clean
root="${GRAFT_REPO_ROOT:-/root/repo}"
case "$1" in
  build) cd "$root" && python tools/probe_variants.py gather ;;
  run)   cd "$root" && LNH_ALLOW_VARIANT=1 bash tools/ab.sh "--skip-bwd" probe0 probe1 probe2 probe3 probe4 probe5 && cat gpurun_out/ab.log ;;
  clean) rm -f "$root"/lidar-nerf_amd/lib/liblidarnerf_hip_probe*.so ;;
  *) echo "usage: $0 build|run|clean"; exit 2 ;;
esac
