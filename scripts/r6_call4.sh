#!/bin/bash
# round 6, call 4: S2 item table + hoisted crossing (no old path), binning parts 2 per CU vs 1 per CU vs 1 part
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
bash scripts/hot_kernels.sh new
HOC_FWD_DBG=$((64*65536)) bash scripts/hot_kernels.sh parts_1_per_cu
HOC_FWD_DBG=$((32*65536)) bash scripts/hot_kernels.sh parts1
bash scripts/hot_kernels.sh c3_new --batch 8 --image-size 480 --image-height 270
HOC_FWD_DBG=$((32*65536)) bash scripts/hot_kernels.sh c3_parts1 --batch 8 --image-size 480 --image-height 270
timeout 900 python -m pytest tests/test_gpu_raster.py -x -q 2>&1 | tail -3
bash scripts/fwd_stage_insts.sh
