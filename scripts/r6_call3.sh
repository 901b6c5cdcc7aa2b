#!/bin/bash
# round 6, call 3: A/B of the forward's S2 hoist and of the barrier-free binning parts (hot path kernels by rocprof)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
bash scripts/hot_kernels.sh new
HOC_FWD_DBG=16384 bash scripts/hot_kernels.sh old_s2
HOC_FWD_DBG=32 bash scripts/hot_kernels.sh parts1
bash scripts/hot_kernels.sh c3_new --batch 8 --image-size 480 --image-height 270
HOC_FWD_DBG=32 bash scripts/hot_kernels.sh c3_parts1 --batch 8 --image-size 480 --image-height 270
timeout 900 python -m pytest tests/test_gpu_raster.py -x -q 2>&1 | tail -3
bash scripts/fwd_stage_insts.sh | head -3
