#!/bin/bash
# One GPU call for the rest of profiles/: grid + MFMA counter passes and the bench lines of the other configurations
# (16 384 rays, bf16 MLP operands, both, the occupancy-grid workload).  profiles/collect.sh writes the default bench line,
# the kernel trace and the FETCH / WRITE_SIZE passes.  Run from the repository root on a GPU box; copy what you want judged
# from gpurun_out/profiles/ to profiles/.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/profiles; mkdir -p $out
timeout 400 bash profiles/collect_counters.sh r02 > gpurun_out/cc.log 2>&1
timeout 250 bash profiles/collect_mfma.sh r02 > gpurun_out/cm.log 2>&1
timeout 200 python bench.py --rays 16384 --no-cpu-baseline > $out/r02_bench_16384.json 2> /dev/null
timeout 200 python bench.py --mlp-dtype bf16 --no-cpu-baseline > $out/r02_bench_bf16.json 2> /dev/null
timeout 200 python bench.py --rays 16384 --mlp-dtype bf16 --no-cpu-baseline > $out/r02_bench_16384_bf16.json 2> /dev/null
timeout 200 python bench.py --workload nerfmvl --no-cpu-baseline > $out/r02_bench_nerfmvl.json 2> /dev/null
ls -la $out
