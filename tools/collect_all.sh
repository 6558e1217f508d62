#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/profiles; mkdir -p $out
timeout 400 bash profiles/collect_counters.sh r02 > gpurun_out/cc.log 2>&1
timeout 250 bash profiles/collect_mfma.sh r02 > gpurun_out/cm.log 2>&1
timeout 200 python bench.py --rays 16384 --no-cpu-baseline > $out/r02_bench_16384.json 2> /dev/null
timeout 200 python bench.py --mlp-dtype bf16 --no-cpu-baseline > $out/r02_bench_bf16.json 2> /dev/null
timeout 200 python bench.py --rays 16384 --mlp-dtype bf16 --no-cpu-baseline > $out/r02_bench_16384_bf16.json 2> /dev/null
timeout 200 python bench.py --workload nerfmvl --no-cpu-baseline > $out/r02_bench_nerfmvl.json 2> /dev/null
ls -la $out
