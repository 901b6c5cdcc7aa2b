#!/bin/bash
# Round profile: rocprofv3 --kernel-trace --stats of the DEFAULT bench command + per-(kernel, grid) split + one
# steady-state step + PMC passes over the per-kernel benchmark.  Usage: scripts/prof_round.sh <tag>   (GPU box)
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_round
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_round -o p -- python $ROOT/bench.py --no-pmc > $OUT/bench_line_profiled.json 2> $OUT/bench.err
f=$(find /tmp/prof_round -name "p_kernel_trace.csv" | head -1)
st=$(find /tmp/prof_round -name "p_kernel_stats.csv" | head -1)
cp "$st" $OUT/bench_kernel_stats.csv
python $ROOT/scripts/stats_by_grid.py "$f" > $OUT/bench_mr_kernels_by_grid.txt
python $ROOT/scripts/step_top_kernels.py "$f" 70 > $OUT/step_kernels.txt
grep "mr::" $OUT/bench_kernel_stats.csv | head -60 > $OUT/bench_mr_kernels.txt
$ROOT/scripts/pmc.sh gpurun_out/prof_$TAG/pmc > $OUT/pmc.log 2>&1
tail -3 $OUT/step_kernels.txt; head -3 $OUT/step_kernels.txt
