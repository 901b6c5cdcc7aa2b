#!/bin/bash
# round 6, call 2: term counter check, hot path launches by ATen op, host profile, forward stage budget, CPU knee
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_call2
mkdir -p $OUT
cd $ROOT
python - > $OUT/def_terms.txt 2>&1 <<'PY'
import json, bench, torch
k = bench.kernel_bench(torch.device("cuda:0"), 64, 256, iters=10, only=("render_backward_full(D+E+F)",)) if hasattr(bench, "kernel_bench") else None
print(json.dumps(k, indent=1))
PY
HOC_ATEN_OPS=1 HOC_HOST_PROFILE=1 python scripts/hot_launches.py > $OUT/hot_launches.txt 2>&1
bash scripts/fwd_stage_insts.sh > /dev/null 2>&1
cp gpurun_out/fwd_stage_insts.txt $OUT/
nproc > $OUT/cpu.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/cpu.txt 2>&1; lscpu | head -20 >> $OUT/cpu.txt
python - > $OUT/cpu_knee.txt 2>&1 <<'PY'
import json, os, bench
print(json.dumps(bench.cpu_baseline_at_the_knee(256, 64, os.cpu_count(), sweep_all=False), indent=1))
PY
timeout 600 python -m pytest tests/test_gpu_bench.py -x -q 2>&1 | tail -5 > $OUT/pytest_bench_tail.txt
cat $OUT/def_terms.txt | tail -30; tail -5 $OUT/pytest_bench_tail.txt
timeout 900 python -m pytest tests/test_gpu_raster.py -x -q -k "binning or bin_counters or tile_list or fused_forward or edge_cases" 2>&1 | tail -5 > $OUT/pytest_raster_tail.txt
tail -3 $OUT/pytest_raster_tail.txt
python bench.py --kernels-only > $OUT/kernels_256.json 2>/dev/null
python bench.py --kernels-only --batch 8 --image-size 480 > $OUT/kernels_480.json 2>/dev/null
python - <<'PY'
import json
for f in ("kernels_256", "kernels_480"):
    d = json.load(open(f"/root/repo/gpurun_out/r6_call2/{f}.json"))
    k = d.get("kernels", d)
    for n in ("render_flow_forward(train outputs,both frames=2B)", "render_vc_forward(train)", "render_forward"):
        if n in k: print(f, n, k[n]["ms"], k[n]["ms_cache_warm"])
PY
