#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/c22
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_graph_step.py tests/test_gpu_trainer.py -m gpu -q 2>&1 | grep -E "^E  |passed|failed" | head -8
{
timeout 400 python scripts/host_timeline.py
timeout 400 python scripts/host_timeline.py --batch 8 --image-size 480 --image-height 270
} > $OUT/host_timeline.txt 2>&1
grep -E "^B=|eager step|graph replay" $OUT/host_timeline.txt
bash scripts/r5_graph.sh c22g 2>&1 | tail -8
