#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_graph_step.py -m gpu -x -q 2>&1 | grep -E "^E  |passed|failed" | head -14
timeout 1200 python -m pytest tests/test_gpu_raster.py tests/test_gpu_chain.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3
HOC_KERNEL_GROUPS="render_backward_full(D+E+F)" timeout 600 python bench.py --kernels-only --batch 32 --image-size 640 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('640', {k[:30]:(v['ms'],v['ms_cache_warm']) for k,v in d.items() if isinstance(v,dict)})"
