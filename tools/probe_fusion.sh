#!/bin/bash
# Encode -> sigma-net fusion question: what does a kernel that walks a point tile through ALL levels inside one
# workgroup lose against the level-major launch?  One library with -DLNH_FWD_LEVEL_LOOP=1 (see k_grid_forward), the forward
# of tools/bench_grid.py timed under rocprofv3 next to the product:
#   bash tools/probe_fusion.sh build     where hipcc is (the variant library travels with the gpurun snapshot)
#   bash tools/probe_fusion.sh run       on a GPU box, from the repository root -> gpurun_out/ab.log
#   bash tools/probe_fusion.sh clean
root="${GRAFT_REPO_ROOT:-/root/repo}"
case "$1" in
  build)
    cd "$root/lidar-nerf_amd" || exit 1
    LNH_VARIANT=fwdloop LNH_EXTRA_FLAGS="-DLNH_FWD_LEVEL_LOOP=1" python build.py > /dev/null 2>&1 || { echo "build failed"; exit 1; }
    cp lib/liblidarnerf_hip.so lib/liblidarnerf_hip_product.so
    ls lib/liblidarnerf_hip_fwdloop.so ;;
  run)
    cd "$root" && bash tools/ab.sh "--skip-bwd" product fwdloop && cat gpurun_out/ab.log ;;
  clean)
    rm -f "$root"/lidar-nerf_amd/lib/liblidarnerf_hip_fwdloop.so "$root"/lidar-nerf_amd/lib/liblidarnerf_hip_product.so
    rm -rf "$root"/lidar-nerf_amd/lib/obj_fwdloop ;;
  *) echo "usage: $0 build|run|clean"; exit 2 ;;
esac
