#!/bin/bash
# A / B of library variants (scripts/build_variant.py) on ONE box: scripts/r6_ab_lib.sh <variant> [<variant> ...]
# ("base" = the library as built); graph-replay device time per hot-path pass + per-kernel durations, metric workload and config 3.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset HOC_LIB_PATH; else export HOC_LIB_PATH=$ROOT/handobjectconsist_amd/variants/lib_$v.so; fi
    echo "metric $v: $(python scripts/hot_only.py --passes 200 2>/dev/null | tail -1)"
    echo "config3 $v: $(python scripts/hot_only.py --batch 8 --image-size 480 --image-height 270 --passes 200 2>/dev/null | tail -1)"
  done
done
for v in "$@"; do
  if [ "$v" = base ]; then unset HOC_LIB_PATH; else export HOC_LIB_PATH=$ROOT/handobjectconsist_amd/variants/lib_$v.so; fi
  bash scripts/hot_kernels.sh $v
  bash scripts/hot_kernels.sh ${v}_c3 --batch 8 --image-size 480 --image-height 270
done
