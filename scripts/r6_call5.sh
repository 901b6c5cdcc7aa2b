#!/bin/bash
# round 6, call 5: the pair step (ABI 8) -- tests, host time, kernels
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 1200 python -m pytest tests/test_gpu_warp.py tests/test_gpu_trainer.py tests/test_gpu_chain.py -x -q 2>&1 | tail -15
bash scripts/hot_kernels.sh step
HOC_PAIR_STEP=0 bash scripts/hot_kernels.sh nodes
python scripts/hot_only.py --passes 200
HOC_PAIR_STEP=0 python scripts/hot_only.py --passes 200
python scripts/hot_only.py --passes 200 --batch 8 --image-size 480 --image-height 270
HOC_PAIR_STEP=0 python scripts/hot_only.py --passes 200 --batch 8 --image-size 480 --image-height 270
python scripts/hot_host_profile.py 200 2>&1 | head -45
