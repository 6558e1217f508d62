#!/bin/bash
# grid parity tests + micro-benchmark + kernel split (scatter / reduce / forward) in one GPU call
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests/test_grid_gpu.py tests/test_zz_properties_gpu.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/t2.log
python tools/bench_grid.py --reps 7 "$@" > gpurun_out/g2.log 2>&1
rm -rf /tmp/kt; rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o r -- python tools/bench_grid.py --reps 5 "$@" > /dev/null 2>&1
python tools/ktrace.py /tmp/kt 2>&1 | head -4 > gpurun_out/g2_kt.log
