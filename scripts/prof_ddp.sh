#!/bin/bash
# rocprofv3 kernel trace of bench.py through the RCCL + DDP path (one rank), summarised by scripts/ddp_overlap.py
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ddp
HOC_FORCE_DDP=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29549 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ddp -o p -- \
  python $ROOT/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-bench --no-stock-trunk > $ROOT/gpurun_out/ddp_bench.json 2> $ROOT/gpurun_out/ddp_bench.err
f=$(find /tmp/prof_ddp -name "p_kernel_trace.csv" | head -1)
python $ROOT/scripts/ddp_overlap.py "$f" | tee $ROOT/gpurun_out/ddp_overlap.txt
python $ROOT/scripts/step_top_kernels.py "$f" 70 > $ROOT/gpurun_out/ddp_step_kernels.txt 2>&1
head -c 600 $ROOT/gpurun_out/ddp_bench.json
