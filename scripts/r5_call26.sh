#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/c26
mkdir -p $OUT
cd $ROOT
G="flow_pair_backward_unit_tiles(train: E scatter of the forward's unit gradient x coefficient,2B)"
for dbg in 0 8; do
  for sz in "--batch 64 --image-size 256" "--batch 32 --image-size 640"; do
    HOC_FLOW_BWD_DBG=$dbg HOC_KERNEL_GROUPS="$G" timeout 600 python bench.py --kernels-only $sz > $OUT/k.json 2>/dev/null
    python -c "
import json; d=json.load(open('$OUT/k.json')); print('dbg=$dbg', '$sz', {k[:30]:(v['ms'],v['ms_cache_warm']) for k,v in d.items() if isinstance(v,dict)})" | tee -a $OUT/ab.txt
  done
done
export HOC_KERNEL_GROUPS="$G" HOC_FLOW_BWD_DBG=8
bash scripts/pmc_kernel.sh c26p "unit_scatter_tiles_kernel" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" $ROOT/bench.py --kernels-only 2>&1 | tail -4 | tee $OUT/pmc.txt
