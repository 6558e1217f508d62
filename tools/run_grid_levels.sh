cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/bench_grid.py --per-level > gpurun_out/grid_levels.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o r -- python tools/bench_grid.py --per-level --reps 2 > /dev/null 2>&1
python tools/ktrace.py /tmp/kt --seq k_grid_bwd > gpurun_out/grid_ktrace.log 2>&1
