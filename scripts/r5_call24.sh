#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/c24
mkdir -p $OUT
cd $ROOT
for mb in 8 16 32; do
  HOC_DDP_BUCKET_MB=$mb timeout 600 python bench.py --gpus 1 --steps 20 --warmup 6 --reducer-ab 3 2>/dev/null | tail -1 | cut -c1-220 | tee -a $OUT/reducer_buckets.txt
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.txt
timeout 1500 python tests/fuzz_parity.py 800 1300000 2>&1 | tail -8 | tee $OUT/fuzz.txt
