#!/bin/bash
# Phase stamps of the binning pass / tile kernel / raster backward on a -DMR_WG_TIMELINE build made ON the GPU box
# (the tree there is a scratch copy).  Usage: scripts/r5_timeline.sh <tag>
TAG=${1:-tl}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export HOC_HIPCC_FLAGS=-DMR_WG_TIMELINE
timeout 600 python handobjectconsist_amd/build.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
timeout 300 python scripts/wg_timeline.py > $OUT/wg_timeline.txt 2>&1
HOC_FWD_FLAGS=$((32 << 24)) timeout 300 python scripts/wg_timeline.py > $OUT/wg_timeline_one_wg_per_image.txt 2>&1
HOC_TL_BATCH=8 HOC_TL_SIZE=480 timeout 300 python scripts/wg_timeline.py > $OUT/wg_timeline_480.txt 2>&1
HOC_TL_BATCH=8 HOC_TL_SIZE=480 HOC_FWD_FLAGS=$((32 << 24)) timeout 300 python scripts/wg_timeline.py > $OUT/wg_timeline_480_one_wg_per_image.txt 2>&1
HOC_TL_BATCH=32 timeout 300 python scripts/wg_timeline.py > $OUT/wg_timeline_b32.txt 2>&1
HOC_TL_BATCH=32 HOC_FWD_FLAGS=$((32 << 24)) timeout 300 python scripts/wg_timeline.py > $OUT/wg_timeline_b32_one_wg_per_image.txt 2>&1
timeout 300 python scripts/bwd_timeline.py > $OUT/bwd_timeline.txt 2>&1
for f in $OUT/wg_timeline*.txt; do echo $f; grep -E '^cold|^binning' $f; done; head -12 $OUT/bwd_timeline.txt
