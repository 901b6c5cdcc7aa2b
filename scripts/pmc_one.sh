#!/bin/bash
# PMC counters of the kernels of one python command, mean per dispatch and kernel:
#   scripts/pmc_one.sh <tag> <kernel substring> <python script + args>
TAG=$1; KSUB=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
  "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH" ; do
  i=$((i+1))
  rm -rf /tmp/pmc1_$i
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc1_$i -o p -- python "$@" > $ROOT/gpurun_out/pmc_${TAG}_$i.log 2>&1
  f=$(find /tmp/pmc1_$i -name "p_counter_collection.csv" | head -1)
  python - "$f" "$KSUB" <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[(r["Kernel_Name"].split("(")[0][:40], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
done
