#!/bin/bash
# Encode -> sigma-net fusion question: what does a kernel that walks a point tile through ALL levels inside one
# workgroup lose against the level-major launch?  The probe build (tools/probe_variants.py fusion: the level loop inside
# k_grid_forward, grid.y = 1) against the product, the forward of tools/bench_grid.py timed under rocprofv3:
#   bash tools/probe_fusion.sh build     where hipcc is (the variant libraries travel with the gpurun snapshot)
#   bash tools/probe_fusion.sh run       on a GPU box, from the repository root -> gpurun_out/ab.log
#   bash tools/probe_fusion.sh clean
root="${GRAFT_REPO_ROOT:-/root/repo}"
case "$1" in
  build) cd "$root" && python tools/probe_variants.py fusion ;;
  run)   cd "$root" && LNH_ALLOW_VARIANT=1 bash tools/ab.sh "--skip-bwd" product fwdloop && cat gpurun_out/ab.log ;;
  clean) rm -f "$root"/lidar-nerf_amd/lib/liblidarnerf_hip_fwdloop.so "$root"/lidar-nerf_amd/lib/liblidarnerf_hip_product.so ;;
  *) echo "usage: $0 build|run|clean"; exit 2 ;;
esac
