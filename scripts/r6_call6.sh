#!/bin/bash
# round 6, call 6: pair step under graph capture; tile kernel with the item table + DPP scans (A/B inside one box via git stash is not possible: compare VALU counts)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
python scripts/hot_only.py --passes 200
HOC_PAIR_STEP=0 python scripts/hot_only.py --passes 200
python scripts/hot_only.py --passes 200 --batch 8 --image-size 480 --image-height 270
HOC_PAIR_STEP=0 python scripts/hot_only.py --passes 200 --batch 8 --image-size 480 --image-height 270
bash scripts/hot_kernels.sh step
bash scripts/fwd_stage_insts.sh
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_warp.py -x -q 2>&1 | tail -3
