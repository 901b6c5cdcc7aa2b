#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/c15
mkdir -p $OUT
cd $ROOT
show() { python -c "
import json; d=json.load(open('$OUT/k.json')); print('$1', {k[:30]:(v['ms'],v['ms_cache_warm']) for k,v in d.items() if isinstance(v,dict) and ('D+E+F' in k or '(E)' in k)})"; }
timeout 900 python bench.py --kernels-only > $OUT/k.json 2>/dev/null; show "all groups, overlap"
HOC_BWD_FLAGS=$((128 << 8)) timeout 900 python bench.py --kernels-only > $OUT/k.json 2>/dev/null; show "all groups, one stream"
HOC_KERNEL_GROUPS="render_backward_full(D+E+F)" timeout 900 python bench.py --kernels-only > $OUT/k.json 2>/dev/null; show "alone, overlap"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_all; rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_all -o p -- python $ROOT/bench.py --kernels-only > /dev/null 2>&1
python $ROOT/scripts/r5_def_trace.py $(find /tmp/prof_all -name "p_kernel_trace.csv" | head -1) | tail -24
rm -rf /tmp/prof_one; HOC_KERNEL_GROUPS="render_backward_full(D+E+F)" rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_one -o p -- python $ROOT/bench.py --kernels-only > /dev/null 2>&1
python $ROOT/scripts/r5_def_trace.py $(find /tmp/prof_one -name "p_kernel_trace.csv" | head -1) | tail -16
