#!/bin/bash
# Round 6, A / B of the launch fusions of the pair step (HOC_PAIR_STEP_FLAGS=2: every stage a launch of its own) on ONE box:
# graph-replay device time per pass + per-kernel durations at the metric workload and at config 3.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for rep in 1 2; do
  for f in 0 2; do
    echo "metric flags=$f: $(HOC_PAIR_STEP_FLAGS=$f python scripts/hot_only.py --passes 200 2>/dev/null | tail -1)"
    echo "config3 flags=$f: $(HOC_PAIR_STEP_FLAGS=$f python scripts/hot_only.py --batch 8 --image-size 480 --image-height 270 --passes 200 2>/dev/null | tail -1)"
  done
done
HOC_PAIR_STEP_FLAGS=0 bash scripts/hot_kernels.sh fused
HOC_PAIR_STEP_FLAGS=2 bash scripts/hot_kernels.sh separate
HOC_PAIR_STEP_FLAGS=0 bash scripts/hot_kernels.sh fused_c3 --batch 8 --image-size 480 --image-height 270
HOC_PAIR_STEP_FLAGS=2 bash scripts/hot_kernels.sh separate_c3 --batch 8 --image-size 480 --image-height 270
