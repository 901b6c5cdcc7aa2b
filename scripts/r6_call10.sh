#!/bin/bash
# round 6, call 10: finalize launch with four rounds per barrier pair (A) vs one (B = variants/lib_dpp.so)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
V=$ROOT/handobjectconsist_amd/variants/lib_dpp.so
timeout 900 python -m pytest tests/test_gpu_warp.py tests/test_gpu_trainer.py -x -q 2>&1 | tail -3
bash scripts/hot_kernels.sh c3_a --batch 8 --image-size 480 --image-height 270
HOC_LIB_PATH=$V bash scripts/hot_kernels.sh c3_b --batch 8 --image-size 480 --image-height 270
bash scripts/hot_kernels.sh c5_a --batch 32 --image-size 640 --image-height 480
HOC_LIB_PATH=$V bash scripts/hot_kernels.sh c5_b --batch 32 --image-size 640 --image-height 480
bash scripts/hot_kernels.sh m_a
HOC_LIB_PATH=$V bash scripts/hot_kernels.sh m_b
