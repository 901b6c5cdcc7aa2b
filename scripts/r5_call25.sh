#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/c25
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_warp.py tests/test_gpu_trainer.py tests/test_gpu_chain.py -m gpu -x -q 2>&1 | tail -3 | tee $OUT/pytest.txt
G="flow_pair_backward_unit_tiles(train: E scatter of the forward's unit gradient x coefficient,2B);flow_pair_backward_tiles(train: pair-loss bwd + epilogue adjoint + E scatter,2B);render_flow_backward(train,E+epilogue adjoint,2B)"
for dbg in 0 4; do
  for sz in "--batch 64 --image-size 256" "--batch 8 --image-size 480" "--batch 32 --image-size 640"; do
    HOC_FLOW_BWD_DBG=$dbg HOC_KERNEL_GROUPS="$G" timeout 600 python bench.py --kernels-only $sz > $OUT/k.json 2>/dev/null
    python -c "
import json; d=json.load(open('$OUT/k.json')); print('dbg=$dbg', '$sz', {k[:30]:(v['ms'],v['ms_cache_warm']) for k,v in d.items() if isinstance(v,dict)})" | tee -a $OUT/ab.txt
  done
done
timeout 300 python scripts/instep.py > $OUT/instep_256.json 2> $OUT/instep_256.err
python -c "
import json; d=json.load(open('$OUT/instep_256.json')); print({k: v['median_us'] for k, v in d.items() if 'scatter' in k or 'flow_pair' in k})" | tee -a $OUT/ab.txt
export HOC_KERNEL_GROUPS="flow_pair_backward_unit_tiles(train: E scatter of the forward's unit gradient x coefficient,2B)"
bash scripts/pmc_kernel.sh c25p "unit_scatter_tiles_kernel" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES" $ROOT/bench.py --kernels-only 2>&1 | tail -8 | tee $OUT/pmc.txt
