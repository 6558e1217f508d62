#!/bin/bash
# One GPU call for profiles/: bash tools/collect_all.sh r03  (from the repository root on a GPU box).
#   profiles/collect.sh            default bench line (with cpu_baseline), kernel trace, FETCH / WRITE_SIZE passes
#   profiles/collect_counters.sh   grid kernel counter groups;  profiles/collect_mfma.sh  MLP kernel counters
#   bench variants: 16 384 rays, bf16 MLP operands, both (BASELINE config 5), the occupancy-grid workload (config 4), the
#   'trained-like' table and the 2x8 patch step of SURVEY 8(d), the data-parallel backward priced on one GPU (--dp-windows),
#   and the CPU baseline with BASELINE.md 3's full protocol.
# Copy what you want judged from gpurun_out/profiles/ to profiles/.
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/profiles; mkdir -p $out
timeout 900 bash profiles/collect.sh $tag > gpurun_out/collect.log 2>&1
timeout 400 bash profiles/collect_counters.sh $tag > gpurun_out/cc.log 2>&1
timeout 250 bash profiles/collect_mfma.sh $tag > gpurun_out/cm.log 2>&1
b() { name=$1; shift; timeout 300 python bench.py "$@" > $out/${tag}_bench_$name.json 2> /dev/null; }
b 16384 --rays 16384 --no-cpu-baseline
b bf16 --mlp-dtype bf16 --no-cpu-baseline
b 16384_bf16 --rays 16384 --mlp-dtype bf16 --no-cpu-baseline
b nerfmvl --workload nerfmvl --no-cpu-baseline
b trained --table trained --no-cpu-baseline
b patch2x8 --patch 2x8 --no-cpu-baseline
b trained_patch2x8 --table trained --patch 2x8 --no-cpu-baseline
b dpwindows --dp-windows --no-cpu-baseline --no-eval
b dpwindows_16384 --dp-windows --rays 16384 --no-cpu-baseline --no-eval
b cpufull --steps 5 --warmup 2 --no-eval --cpu-baseline-full
ls -la $out
