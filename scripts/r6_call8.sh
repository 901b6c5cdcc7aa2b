#!/bin/bash
# round 6, call 8: per-face set-up in the records (A = this build) vs the build before it (B = variants/lib_dpp.so), one box
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
V=$ROOT/handobjectconsist_amd/variants/lib_dpp.so
timeout 900 python -m pytest tests/test_gpu_raster.py -x -q 2>&1 | tail -3
bash scripts/hot_kernels.sh rec
HOC_LIB_PATH=$V bash scripts/hot_kernels.sh dpp
bash scripts/hot_kernels.sh rec2
HOC_LIB_PATH=$V bash scripts/hot_kernels.sh dpp2
bash scripts/hot_kernels.sh c3_rec --batch 8 --image-size 480 --image-height 270
HOC_LIB_PATH=$V bash scripts/hot_kernels.sh c3_dpp --batch 8 --image-size 480 --image-height 270
bash scripts/fwd_stage_insts.sh
python bench.py --kernels-only > gpurun_out/kernels_rec.json 2>/dev/null
HOC_LIB_PATH=$V python bench.py --kernels-only > gpurun_out/kernels_dpp.json 2>/dev/null
python - <<'PY'
import json
for f in ("kernels_rec", "kernels_dpp"):
    d = json.load(open(f"/root/repo/gpurun_out/{f}.json"))
    k = d.get("kernels", d)
    for n in ("render_flow_forward(train outputs,both frames=2B)", "render_vc_forward(train)", "render_vc_forward(train,both frames=2B)", "render_forward"):
        if n in k: print(f, n, k[n]["ms"], k[n]["ms_cache_warm"])
PY
timeout 1200 python -m pytest tests/test_gpu_warp.py tests/test_gpu_chain.py tests/test_gpu_trainer.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3
