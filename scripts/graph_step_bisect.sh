#!/bin/bash
# Round 6, time-boxed (<= 30 GPU-minutes): does a hipGraph-replayed train step return the eager step's gradients when MIOpen's
# atomic split-K weight-gradient solvers are denied / the capture warms up longer?  Output: gpurun_out/graph_bisect.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
run() { echo "=== $1"; A=$2; shift 2; env "$@" timeout 280 python scripts/graph_step_bisect.py --replays 50 $A 2>&1 | grep -v Warning | tail -6; }
{
run "baseline (warm 1)" "--warm 1" X=1
run "warm 3" "--warm 3" X=1
run "warm 3, no implicit-GEMM asm wrw (gtc xdlops nhwc / nchw)" "--warm 3" MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_WRW_GTC_XDLOPS_NHWC=0 MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_WRW_GTC_XDLOPS=0
run "warm 3, no implicit GEMM at all" "--warm 3" MIOPEN_DEBUG_CONV_IMPLICIT_GEMM=0
run "warm 3, GEMM wrw only (no direct / winograd / implicit gemm / fft)" "--warm 3" MIOPEN_DEBUG_CONV_IMPLICIT_GEMM=0 MIOPEN_DEBUG_CONV_DIRECT=0 MIOPEN_DEBUG_CONV_WINOGRAD=0 MIOPEN_DEBUG_CONV_FFT=0
run "warm 3, solver search on (cudnn.benchmark)" "--warm 3 --benchmark 1" X=1
} > $OUT/graph_bisect.txt 2>&1
cat $OUT/graph_bisect.txt
