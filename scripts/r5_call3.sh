#!/bin/bash
# scatter over work lists + kernel D without LDS float atomics: parity tests, kernel benches, phase stamps
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/c3
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_gpu_warp.py tests/test_gpu_trainer.py tests/test_gpu_raster.py -m gpu -x -q -k "scatter_work or fused_pair_node or metric_workload or fused_backward or strip or few_pixels or sparse_warp" 2>&1 | tail -30 > $OUT/pytest.txt
tail -3 $OUT/pytest.txt
G="flow_pair_backward_unit_tiles(train: E scatter of the forward's unit gradient x coefficient,2B);flow_pair_forward_grad_tiles(train: occlusion + epilogue + pair loss + unit gradient, sparse);render_backward_full(D+E+F);flow_pair_backward_tiles(train: pair-loss bwd + epilogue adjoint + E scatter,2B);render_flow_backward(train,E+epilogue adjoint,2B)"
HOC_KERNEL_GROUPS="$G" timeout 600 python bench.py --kernels-only > $OUT/kernels_256.json 2> $OUT/kernels_256.err
HOC_KERNEL_GROUPS="$G" timeout 600 python bench.py --kernels-only --batch 8 --image-size 480 > $OUT/kernels_480.json 2> $OUT/kernels_480.err
HOC_KERNEL_GROUPS="$G" timeout 600 python bench.py --kernels-only --batch 32 --image-size 640 > $OUT/kernels_640.json 2> $OUT/kernels_640.err
HOC_BWD_FLAGS=$((16 << 8)) HOC_KERNEL_GROUPS="render_backward_full(D+E+F)" timeout 600 python bench.py --kernels-only > $OUT/kernels_256_lds_float_atomics.json 2>/dev/null
HOC_FLOW_BWD_DBG=$((1 << 7)) HOC_KERNEL_GROUPS="$G" timeout 600 python bench.py --kernels-only > $OUT/kernels_256_listing_form.json 2>/dev/null
timeout 300 python scripts/instep.py > $OUT/instep_256.json 2> $OUT/instep_256.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/kernels_*.json")) :
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], {k[:34]: (v.get("ms"), v.get("ms_cache_warm")) for k, v in d.items() if isinstance(v, dict)})
    except Exception as e:
        print(f, "unreadable", e)
try:
    d = json.load(open("$OUT/instep_256.json")); print({k: v["median_us"] for k, v in d.items()})
except Exception as e: print("instep", e)
PY
export HOC_HIPCC_FLAGS=-DMR_WG_TIMELINE
timeout 600 python handobjectconsist_amd/build.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
timeout 300 python scripts/bwd_timeline.py > $OUT/bwd_timeline.txt 2>&1
tail -12 $OUT/bwd_timeline.txt
