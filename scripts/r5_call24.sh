#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for mb in 8 16 32; do
  HOC_DDP_BUCKET_MB=$mb timeout 600 python bench.py --gpus 1 --steps 20 --warmup 6 --reducer-ab 3 2>/dev/null | tail -1 | cut -c1-220
done
timeout 1500 python tests/fuzz_parity.py 1200 1300000 2>&1 | tail -6
