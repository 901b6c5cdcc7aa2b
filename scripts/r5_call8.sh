#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/c8
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_chain.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -5 > $OUT/pytest.txt
tail -2 $OUT/pytest.txt
export HOC_KERNEL_GROUPS="render_backward_full(D+E+F);render_backward_train(E)"
for f in 0 128; do
  export HOC_BWD_FLAGS=$((f << 8))
  for sz in "--batch 64 --image-size 256" "--batch 8 --image-size 480" "--batch 32 --image-size 640"; do
    timeout 600 python bench.py --kernels-only $sz > $OUT/k.json 2>/dev/null
    python -c "
import json; d=json.load(open('$OUT/k.json')); print('flags>>8=$f', '$sz', {k[:30]:(v['ms'],v['ms_cache_warm']) for k,v in d.items() if isinstance(v,dict)})"
  done
done
export HOC_BWD_FLAGS=0
bash scripts/prof_kernels.sh c8 $ROOT/bench.py --kernels-only > /dev/null 2>&1
grep -E "gather_kernel|pixel_map_strip|compact|mark|strip_list" $ROOT/gpurun_out/prof_c8_by_grid.txt
