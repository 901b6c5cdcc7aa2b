#!/bin/bash
# kernel trace of steady-state steps: graph replay vs eager  ->  gpurun_out/<tag>/step_{graph,eager}_{kernels,sequence}.txt
TAG=${1:-tr}
ARGS=${2:-}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in graph eager; do
  rm -rf /tmp/prof_$mode
  extra=""; [ $mode = eager ] && extra="--eager-step"
  HOC_CUDNN_BENCHMARK=1 timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$mode -o p -- python $ROOT/bench.py --step-only --steps 8 --warmup 5 $ARGS $extra > $OUT/step_$mode.json 2> $OUT/step_$mode.err
  f=$(find /tmp/prof_$mode -name "p_kernel_trace.csv" | head -1)
  python $ROOT/scripts/step_top_kernels.py "$f" 30 $OUT/step_${mode}_sequence.txt > $OUT/step_${mode}_kernels.txt 2>> $OUT/step_$mode.err
  head -2 $OUT/step_${mode}_kernels.txt; cat $OUT/step_$mode.json
done
