#!/bin/bash
# PMC passes over the per-kernel benchmark (one rocprofv3 run per counter group: --pmc must not
# be combined with trace domains other than --kernel-trace).  Usage: scripts/pmc.sh <outdir> [bench args]
set -u
OUT=${1:-gpurun_out/pmc}; shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS_ATOMIC" \
  "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc_$i -o p -- python $ROOT/bench.py --kernels-only --kernel-iters 3 "$@" > "$ROOT/$OUT/pass$i.log" 2>&1
  cp /tmp/pmc_$i/p_counter_collection.csv "$ROOT/$OUT/pass${i}_counters.csv" 2>/dev/null || echo "pass $i: no counter csv"
done
python $ROOT/scripts/pmc_summary.py "$ROOT/$OUT" --json "$ROOT/$OUT/traffic.json" > "$ROOT/$OUT/summary.txt"
tail -5 "$ROOT/$OUT/summary.txt"
