#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/c12
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_graph_step.py -m gpu -q 2>&1 | tail -3
COMMON="--no-kernel-bench --no-cpu-baseline --no-stock-trunk --batch 32 --image-size 640 --image-height 480 --encoder-dtype bf16 --steps 30"
for k in 1 2 3; do
  timeout 400 python bench.py $COMMON > $OUT/g640_$k.json 2> $OUT/g640_$k.err; echo "work=1 run $k rc=$? $(grep -c 'Loss became nan' $OUT/g640_$k.err) $(cut -c1-120 $OUT/g640_$k.json)"
done
for k in 1 2; do
  HOC_SCATTER_WORK=0 timeout 400 python bench.py $COMMON > $OUT/g640_nw_$k.json 2> $OUT/g640_nw_$k.err; echo "work=0 run $k rc=$? $(grep -c 'Loss became nan' $OUT/g640_nw_$k.err) $(cut -c1-120 $OUT/g640_nw_$k.json)"
done
export HOC_KERNEL_GROUPS="render_flow_forward(train outputs,both frames=2B)"
for b in 0 1792 2048 3584 4096; do
  if [ $b = 0 ]; then unset HOC_TILE_BOUND; else export HOC_TILE_BOUND=$b; fi
  timeout 300 python bench.py --kernels-only > $OUT/k.json 2>/dev/null
  python -c "
import json; d=json.load(open('$OUT/k.json')); print('tile bound $b', {k[:30]:(v['ms'],v['ms_cache_warm']) for k,v in d.items() if isinstance(v,dict)})"
done
