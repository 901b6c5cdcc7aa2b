#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/c13
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_graph_step.py -m gpu -q -x 2>&1 | grep -E "^E  |^tests/|passed|failed" | head -12
COMMON="--no-kernel-bench --no-cpu-baseline --no-stock-trunk --batch 32 --image-size 640 --image-height 480 --encoder-dtype bf16 --steps 30"
for k in 1 2 3; do
  timeout 400 python bench.py $COMMON > $OUT/g640_$k.json 2> $OUT/g640_$k.err; echo "run $k rc=$?"; grep "\[bench\] timed step" $OUT/g640_$k.err
done
HOC_TUNABLEOP=0 timeout 400 python bench.py $COMMON > $OUT/g640_nt.json 2> $OUT/g640_nt.err; echo "no tunableop rc=$?"; grep "\[bench\] timed step" $OUT/g640_nt.err
timeout 400 python bench.py $COMMON --encoder-dtype f32 > $OUT/g640_f32.json 2> $OUT/g640_f32.err; echo "f32 rc=$?"; grep "\[bench\] timed step" $OUT/g640_f32.err
