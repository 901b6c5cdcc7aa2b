#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/c6
mkdir -p $OUT
cd $ROOT
export HOC_KERNEL_GROUPS="render_backward_full(D+E+F);render_backward_train(E)"
run() {  # tag, hipcc flags
  export HOC_HIPCC_FLAGS="$2"
  timeout 600 python handobjectconsist_amd/build.py > $OUT/build_$1.log 2>&1 || { tail -5 $OUT/build_$1.log; return; }
  timeout 900 python -m pytest tests/test_gpu_raster.py -m gpu -x -q -k "fused_backward or strip or few_pixels or compat" 2>&1 | tail -3 > $OUT/pytest_$1.txt
  timeout 600 python bench.py --kernels-only > $OUT/kernels_$1.json 2>/dev/null
  bash scripts/prof_kernels.sh c6$1 $ROOT/bench.py --kernels-only > /dev/null 2>&1
  grep -E "gather_kernel|pixel_map_strip|compact" $ROOT/gpurun_out/prof_c6$1_by_grid.txt > $OUT/by_grid_$1.txt
  echo "== $1 ($2)"; tail -1 $OUT/pytest_$1.txt; cat $OUT/by_grid_$1.txt
  python -c "
import json; d=json.load(open('$OUT/kernels_$1.json')); print({k[:30]:(v['ms'],v['ms_cache_warm']) for k,v in d.items() if isinstance(v,dict)})"
}
run ggl8 ""
run ggl4 "-DMR_GGL=4"
run ggl8w4 "-DMR_GATHER_WPE=4"
run ggl4w4 "-DMR_GGL=4 -DMR_GATHER_WPE=4"
run ggl4w5 "-DMR_GGL=4 -DMR_GATHER_WPE=5"
