#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/c9
mkdir -p $OUT
cd $ROOT
export HOC_KERNEL_GROUPS="render_backward_full(D+E+F);render_backward_train(E)"
run() {
  export HOC_HIPCC_FLAGS="$2"
  timeout 600 python handobjectconsist_amd/build.py > $OUT/build_$1.log 2>&1 || { tail -5 $OUT/build_$1.log; return; }
  timeout 900 python -m pytest tests/test_gpu_raster.py -m gpu -x -q -k "fused_backward or strip or few_pixels" 2>&1 | tail -1
  for sz in "--batch 64 --image-size 256" "--batch 8 --image-size 480"; do
    timeout 600 python bench.py --kernels-only $sz > $OUT/k.json 2>/dev/null
    python -c "
import json; d=json.load(open('$OUT/k.json')); print('$1', '$sz', {k[:30]:(v['ms'],v['ms_cache_warm']) for k,v in d.items() if isinstance(v,dict)})"
  done
  bash scripts/prof_kernels.sh c9$1 $ROOT/bench.py --kernels-only > /dev/null 2>&1
  grep -E "compact|mark" $ROOT/gpurun_out/prof_c9$1_by_grid.txt
}
run per4 ""
run per2 "-DMR_CO_PER=2"
run per1 "-DMR_CO_PER=1"
