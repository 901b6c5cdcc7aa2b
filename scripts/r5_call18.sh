#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/c18
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_full; HOC_KERNEL_GROUPS_FULL="render_backward_full(D+E+F)" rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_full -o p -- python $ROOT/bench.py --no-cpu-baseline --no-stock-trunk --no-pmc --steps 4 --warmup 4 > $OUT/b.json 2>/dev/null
python $ROOT/scripts/r5_def_trace.py $(find /tmp/prof_full -name "p_kernel_trace.csv" | head -1) | tail -40
cd $ROOT
HOC_TUNABLEOP=0 timeout 900 python bench.py --no-cpu-baseline --no-stock-trunk --no-pmc --steps 4 --warmup 4 > $OUT/b2.json 2>/dev/null
python - <<PY
import json
d=json.loads([l for l in open("$OUT/b2.json") if l.startswith("{")][-1])
print("no tunableop:", {k[:28]:(v["ms"],v["ms_cache_warm"]) for k,v in d["kernels"].items() if "D+E+F" in k})
PY
