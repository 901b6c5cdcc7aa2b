#!/bin/bash
# One steady-state training step of the default configuration by kernel (rocprofv3 kernel trace of
# bench.py without the stock-trunk / per-kernel legs).  Usage: scripts/prof_step.sh <tag> [env assignments]
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_step
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_step -o p -- python $ROOT/bench.py --steps 8 --warmup 5 --no-pmc --no-stock-trunk --no-kernel-bench --no-cpu-baseline > $OUT/step_bench_line.json 2> $OUT/step_bench.err
f=$(find /tmp/prof_step -name "p_kernel_trace.csv" | head -1)
python $ROOT/scripts/step_top_kernels.py "$f" 80 $OUT/step_sequence.txt > $OUT/step_kernels.txt
python $ROOT/scripts/hot_launches.py > $OUT/hot_path_launches.txt 2>&1
head -4 $OUT/step_kernels.txt; cut -c1-300 $OUT/step_bench_line.json | head -2
