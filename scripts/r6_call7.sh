#!/bin/bash
# round 6, call 7: mean inside the finalize launch; tile kernel A/B (item table + DPP scans vs round 5's) on one box
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_warp.py tests/test_gpu_trainer.py -x -q 2>&1 | tail -3
python scripts/hot_only.py --passes 300
python scripts/hot_only.py --passes 300 --batch 8 --image-size 480 --image-height 270
bash scripts/hot_kernels.sh new
HOC_LIB_PATH=$ROOT/handobjectconsist_amd/variants/lib_tile_r5.so bash scripts/hot_kernels.sh tile_r5
bash scripts/hot_kernels.sh new2
HOC_LIB_PATH=$ROOT/handobjectconsist_amd/variants/lib_tile_r5.so bash scripts/hot_kernels.sh tile_r5_2
bash scripts/hot_kernels.sh c3_new --batch 8 --image-size 480 --image-height 270
HOC_LIB_PATH=$ROOT/handobjectconsist_amd/variants/lib_tile_r5.so bash scripts/hot_kernels.sh c3_tile_r5 --batch 8 --image-size 480 --image-height 270
