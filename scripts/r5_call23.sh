#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/c23
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_graph_step.py -m gpu -q -s 2>&1 | grep -E "^E  |passed|failed|replayed vs eager" | head -8
export HOC_KERNEL_GROUPS="render_backward_full(D+E+F);render_backward_train(E)"
for big in 256 1024 8192; do
  export HOC_HIPCC_FLAGS="-DMR_GATHER_BIG=$big"
  timeout 600 python handobjectconsist_amd/build.py > $OUT/build.log 2>&1 || { tail -5 $OUT/build.log; continue; }
  timeout 600 python -m pytest tests/test_gpu_raster.py -m gpu -x -q -k "fused_backward or big or strip" 2>&1 | tail -1
  for sz in "--batch 64 --image-size 256" "--batch 32 --image-size 640"; do
    timeout 600 python bench.py --kernels-only $sz > $OUT/k.json 2>/dev/null
    python -c "
import json; d=json.load(open('$OUT/k.json')); print('big>$big', '$sz', {k[:30]:(v['ms'],v['ms_cache_warm']) for k,v in d.items() if isinstance(v,dict)})"
  done
done
