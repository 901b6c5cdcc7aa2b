#!/bin/bash
# round 6, call 12: power-of-two pixel centres by arithmetic in S2 (variants/lib_p2.so) vs the table; host timeline, repeated
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_call12
mkdir -p $OUT
cd $ROOT
V=$ROOT/handobjectconsist_amd/variants/lib_p2.so
bash scripts/hot_kernels.sh tab
HOC_LIB_PATH=$V bash scripts/hot_kernels.sh p2
bash scripts/hot_kernels.sh tab_2
HOC_LIB_PATH=$V bash scripts/hot_kernels.sh p2_2
HOC_LIB_PATH=$V timeout 600 python -m pytest tests/test_gpu_raster.py -x -q 2>&1 | tail -2
for i in 1 2 3; do
  python scripts/host_timeline.py --no-graph 2>/dev/null | grep "eager step" | sed 's/^/step /' >> $OUT/host_timeline.txt
  HOC_PAIR_STEP=0 python scripts/host_timeline.py --no-graph 2>/dev/null | grep "eager step" | sed 's/^/[HOC_PAIR_STEP=0] /' >> $OUT/host_timeline.txt
done
cat $OUT/host_timeline.txt
