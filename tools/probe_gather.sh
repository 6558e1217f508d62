#!/bin/bash
# Gather cost model of the encode forward.  One library per LNH_GATHER_PROBE value (see the comment in
# k_grid_forward), the forward of tools/bench_grid.py timed under rocprofv3:
#   bash tools/probe_gather.sh build     where hipcc is (the variant libraries travel with the gpurun snapshot)
#   bash tools/probe_gather.sh run       on a GPU box, from the repository root -> gpurun_out/ab.log
#   bash tools/probe_gather.sh clean
root="${GRAFT_REPO_ROOT:-/root/repo}"
case "$1" in
  build)
    cd "$root/lidar-nerf_amd" || exit 1
    for p in 0 1 2 3 4 5; do
      LNH_VARIANT=probe$p LNH_EXTRA_FLAGS="-DLNH_GATHER_PROBE=$p" python build.py > /dev/null 2>&1 || { echo "build of probe $p failed"; exit 1; }
    done
    ls lib/liblidarnerf_hip_probe*.so ;;
  run)
    cd "$root" && bash tools/ab.sh "--skip-bwd" probe0 probe1 probe2 probe3 probe4 probe5 && cat gpurun_out/ab.log ;;
  clean)
    rm -f "$root"/lidar-nerf_amd/lib/liblidarnerf_hip_probe*.so; rm -rf "$root"/lidar-nerf_amd/lib/obj_probe* ;;
  *) echo "usage: $0 build|run|clean"; exit 2 ;;
esac
