#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/c16
mkdir -p $OUT
cd $ROOT
show() { python - <<PY
import json
d=json.loads([l for l in open("$OUT/b.json") if l.startswith("{")][-1])
print("$1", d["value"], {k[:28]:(v["ms"],v["ms_cache_warm"]) for k,v in d["kernels"].items() if "D+E+F" in k})
PY
}
GPU_MAX_HW_QUEUES=8 timeout 900 python bench.py --no-cpu-baseline --no-stock-trunk --no-pmc > $OUT/b.json 2> $OUT/b8.err; show "GPU_MAX_HW_QUEUES=8"
timeout 900 python bench.py --no-cpu-baseline --no-stock-trunk --no-pmc > $OUT/b.json 2> $OUT/b4.err; show "default queues"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_full; rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_full -o p -- python $ROOT/bench.py --no-cpu-baseline --no-stock-trunk --no-pmc --steps 4 --warmup 4 > /dev/null 2>&1
python $ROOT/scripts/r5_def_trace.py $(find /tmp/prof_full -name "p_kernel_trace.csv" | head -1) | tail -14
