#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/c10
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_graph_step.py -m gpu -x -q 2>&1 | tail -2
{
timeout 400 python scripts/host_timeline.py
timeout 400 python scripts/host_timeline.py --batch 8 --image-size 480 --image-height 270
timeout 400 python scripts/host_timeline.py --batch 32 --image-size 640 --image-height 480 --encoder-dtype bf16
} > $OUT/host_timeline.txt 2>&1
grep -E "^B=|eager step|graph replay" $OUT/host_timeline.txt
bash scripts/r5_graph.sh c10g 2>&1 | tail -8
