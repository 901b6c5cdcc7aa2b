#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_graph_step.py -m gpu -x -q 2>&1 | grep -E "Error|error|assert|FAILED|passed|failed|ACTUAL|DESIRED|Mismatch|differ" | head -20
for cfg in "32 640 480 bf16 1 40 1" "32 640 480 bf16 0 40 1" "32 640 480 f32 1 40 1" "32 640 480 bf16 1 40 0" "8 480 270 bf16 1 40 1" "64 256 256 bf16 1 40 1"; do
  timeout 300 python scripts/r5_debug_nan.py $cfg 2>&1 | tail -1
done
