#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/c27
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_warp.py tests/test_gpu_trainer.py -m gpu -x -q -k "scatter_work or fused_pair_node or metric_workload or sparse_warp" 2>&1 | tail -2 | tee $OUT/pytest.txt
G="flow_pair_backward_unit_tiles(train: E scatter of the forward's unit gradient x coefficient,2B)"
for sz in "--batch 64 --image-size 256" "--batch 8 --image-size 480" "--batch 32 --image-size 640"; do
  HOC_KERNEL_GROUPS="$G" timeout 600 python bench.py --kernels-only $sz > $OUT/k.json 2>/dev/null
  python -c "
import json; d=json.load(open('$OUT/k.json')); print('$sz', {k[:30]:(v['ms'],v['ms_cache_warm']) for k,v in d.items() if isinstance(v,dict)})" | tee -a $OUT/ab.txt
done
timeout 300 python scripts/instep.py > $OUT/instep_256.json 2> $OUT/instep_256.err
python -c "
import json; d=json.load(open('$OUT/instep_256.json')); print({k: v['median_us'] for k, v in d.items() if 'scatter' in k or 'flow_pair' in k})" | tee -a $OUT/ab.txt
