#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/c19
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_gpu_raster.py tests/test_gpu_chain.py tests/test_gpu_fuzz.py tests/test_gpu_graph_step.py -m gpu -x -q 2>&1 | tail -4
export HOC_KERNEL_GROUPS="render_backward_full(D+E+F);render_backward_train(E)"
for f in 0 128; do
  export HOC_BWD_FLAGS=$((f << 8))
  for sz in "--batch 64 --image-size 256" "--batch 8 --image-size 480" "--batch 32 --image-size 640"; do
    timeout 600 python bench.py --kernels-only $sz > $OUT/k.json 2>/dev/null
    python -c "
import json; d=json.load(open('$OUT/k.json')); print('flags>>8=$f', '$sz', {k[:30]:(v['ms'],v['ms_cache_warm']) for k,v in d.items() if isinstance(v,dict)})"
  done
done
unset HOC_BWD_FLAGS HOC_KERNEL_GROUPS
timeout 900 python bench.py --no-cpu-baseline --no-stock-trunk --no-pmc > $OUT/b.json 2> $OUT/b.err
python - <<PY
import json
d=json.loads([l for l in open("$OUT/b.json") if l.startswith("{")][-1])
print("full bench:", d["value"], {k[:28]:(v["ms"],v["ms_cache_warm"]) for k,v in d["kernels"].items() if "D+E+F" in k or "(E)" in k})
PY
cd /tmp && export TMPDIR=/tmp
HOC_KERNEL_GROUPS="render_backward_full(D+E+F)" bash $ROOT/scripts/prof_kernels.sh c19 $ROOT/bench.py --kernels-only > /dev/null 2>&1
grep -E "strip_gather|gather_kernel|pixel_map_strip|compact|mark|strip_list" $ROOT/gpurun_out/prof_c19_by_grid.txt
