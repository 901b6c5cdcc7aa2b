#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/c14
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_graph_step.py -m gpu -x -q 2>&1 | grep -E "^E  |passed|failed" | head -12
timeout 1500 python -m pytest tests/test_gpu_warp.py tests/test_gpu_trainer.py tests/test_gpu_graph_step.py tests/test_gpu_chain.py -m gpu -x -q 2>&1 | tail -8 > $OUT/pytest.txt
tail -3 $OUT/pytest.txt
G="flow_pair_backward_unit_tiles(train: E scatter of the forward's unit gradient x coefficient,2B);flow_pair_forward_grad_tiles(train: occlusion + epilogue + pair loss + unit gradient, sparse);flow_pair_forward_tiles(occlusion + epilogue + pair loss, sparse)"
HOC_KERNEL_GROUPS="$G" timeout 600 python bench.py --kernels-only > $OUT/kernels_256.json 2> $OUT/kernels_256.err
HOC_KERNEL_GROUPS="$G" timeout 600 python bench.py --kernels-only --batch 8 --image-size 480 > $OUT/kernels_480.json 2> $OUT/kernels_480.err
HOC_KERNEL_GROUPS="$G" timeout 600 python bench.py --kernels-only --batch 32 --image-size 640 > $OUT/kernels_640.json 2> $OUT/kernels_640.err
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
    d = json.load(open("$OUT/instep_256.json")); print({k: v["median_us"] for k, v in d.items() if "scatter" in k or "finalize" in k or "flow_pair" in k})
except Exception as e: print("instep", e)
PY
export HOC_HIPCC_FLAGS=-DMR_WG_TIMELINE
timeout 600 python handobjectconsist_amd/build.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
timeout 300 python scripts/bwd_timeline.py > $OUT/bwd_timeline.txt 2>&1
head -16 $OUT/bwd_timeline.txt
unset HOC_HIPCC_FLAGS
timeout 600 python handobjectconsist_amd/build.py > $OUT/build2.log 2>&1
timeout 900 python bench.py --no-cpu-baseline > $OUT/bench_full.json 2> $OUT/bench_full.err
python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench_full.json") if l.startswith("{")][-1])
print("full bench:", d["value"], d["ms_per_step"], d["hot_path_ms"], {k[:28]:(v["ms"],v["ms_cache_warm"]) for k,v in d["kernels"].items() if "D+E+F" in k or "unit" in k or "grad_tiles" in k})
print({k:d["roofline"].get(k) for k in ("frac","frac_cache_warm","frac_in_step","in_step_us")})
PY
