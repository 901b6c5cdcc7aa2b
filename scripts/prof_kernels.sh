#!/bin/bash
# rocprofv3 kernel trace of a command, summarised per (kernel, grid):  scripts/prof_kernels.sh <tag> <python script + args>
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$TAG -o p -- python "$@" > $ROOT/gpurun_out/prof_$TAG.log 2>&1
f=$(find /tmp/prof_$TAG -name "p_kernel_trace.csv" | head -1)
python $ROOT/scripts/stats_by_grid.py "$f" | tee $ROOT/gpurun_out/prof_${TAG}_by_grid.txt
