#!/bin/bash
cd $GRAFT_REPO_ROOT
for fl in 0 2048 256; do
  echo "== flags $fl"
  HOC_KERNEL_GROUPS="render_backward_full(D+E+F)" HOC_BWD_FLAGS=$fl timeout 120 bash scripts/prof_kernels.sh dstrip$fl $GRAFT_REPO_ROOT/bench.py --kernels-only --kernel-iters 30 | grep "strip"
done
