#!/bin/bash
# round 6, call 11: tile kernel at 8 waves per SIMD (64 registers, 12 spilled) vs 7; host timeline of the step; kernels at 480 / 640
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_call11
mkdir -p $OUT
cd $ROOT
V=$ROOT/handobjectconsist_amd/variants/lib_tile8.so
bash scripts/hot_kernels.sh w7
HOC_LIB_PATH=$V bash scripts/hot_kernels.sh w8
bash scripts/hot_kernels.sh w7_2
HOC_LIB_PATH=$V bash scripts/hot_kernels.sh w8_2
bash scripts/hot_kernels.sh c3_w7 --batch 8 --image-size 480 --image-height 270
HOC_LIB_PATH=$V bash scripts/hot_kernels.sh c3_w8 --batch 8 --image-size 480 --image-height 270
for cfg in "--batch 64 --image-size 256" "--batch 8 --image-size 480 --image-height 270" "--batch 32 --image-size 640 --image-height 480 --encoder-dtype bf16"; do
  python scripts/host_timeline.py --no-graph $cfg 2>/dev/null | grep -v "^$" >> $OUT/host_timeline.txt
  HOC_PAIR_STEP=0 python scripts/host_timeline.py --no-graph $cfg 2>/dev/null | sed 's/^/    [HOC_PAIR_STEP=0] /' >> $OUT/host_timeline.txt
done
cat $OUT/host_timeline.txt
python bench.py --kernels-only --batch 8 --image-size 480 > $OUT/r06_kernels_480.json 2>/dev/null
python bench.py --kernels-only --batch 32 --image-size 640 > $OUT/r06_kernels_640.json 2>/dev/null
python - <<'PY'
import json
for f in ("r06_kernels_480", "r06_kernels_640"):
    d = json.load(open(f"/root/repo/gpurun_out/r6_call11/{f}.json"))
    k = d.get("kernels", d)
    for n, v in k.items():
        if any(w in n for w in ("flow_forward", "unit_tiles", "grad_tiles", "D+E+F")): print(f, n[:60], v["ms"], v["ms_cache_warm"], v.get("frac_hbm_peak"), v.get("frac_compulsory"))
PY
