#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/c20
mkdir -p $OUT
cd $ROOT
export HOC_KERNEL_GROUPS="render_backward_full(D+E+F)"
for f in 0 256 512 128; do
  export HOC_BWD_FLAGS=$((f << 8))
  for sz in "--batch 64 --image-size 256" "--batch 8 --image-size 480"; do
    timeout 600 python bench.py --kernels-only $sz > $OUT/k.json 2>/dev/null
    python -c "
import json; d=json.load(open('$OUT/k.json')); print('flags>>8=$f', '$sz', {k[:30]:(v['ms'],v['ms_cache_warm']) for k,v in d.items() if isinstance(v,dict)})"
  done
done
