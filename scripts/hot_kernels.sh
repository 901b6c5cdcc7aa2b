#!/bin/bash
# Per-kernel durations of the render + warp hot path (scripts/hot_only.py under rocprofv3 --kernel-trace), eager + replayed passes.
# Usage: scripts/hot_kernels.sh <label> [hot_only.py arguments]   (environment, e.g. HOC_FWD_DBG=..., is inherited)
LABEL=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/hot_k
rocprofv3 --kernel-trace --output-format csv -d /tmp/hot_k -o p -- python $ROOT/scripts/hot_only.py "$@" > $OUT/hot_kernels_$LABEL.json 2> /tmp/hot_k.err
f=$(find /tmp/hot_k -name "p_kernel_trace.csv" | head -1)
python $ROOT/scripts/stats_by_grid.py "$f" > $OUT/hot_kernels_$LABEL.txt
echo "== $LABEL: $(cat $OUT/hot_kernels_$LABEL.json)"
grep -v "^kernel" $OUT/hot_kernels_$LABEL.txt | awk '{printf "%-50s %6s %8s %8s\n", $1" "$2, $(NF-4), $(NF-3), $(NF-1)}' | head -20
