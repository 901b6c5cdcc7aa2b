#!/bin/bash
# PMC counters of one kernel over a command:  scripts/pmc_kernel.sh <tag> <kernel name substring> "<counters>" <python script + args>
# (counters in their own pass, with the kernel trace only; per-dispatch values averaged over the launches after the first)
TAG=$1; KERNEL=$2; COUNTERS=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$TAG
rocprofv3 --pmc $COUNTERS --kernel-trace --output-format csv -d /tmp/pmc_$TAG -o p -- python "$@" > $ROOT/gpurun_out/pmc_$TAG.log 2>&1
f=$(find /tmp/pmc_$TAG -name "p_counter_collection.csv" | head -1)
python - "$f" "$KERNEL" <<'PY' | tee $ROOT/gpurun_out/pmc_$TAG.txt
import csv, sys
from collections import defaultdict
rows = defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = rows[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(rows)[1:]
print(sys.argv[2], "dispatches", len(ids))
for c in sorted({c for d in ids for c in rows[d]}):
    print("  %-28s %14.3f M per launch" % (c, sum(rows[d].get(c, 0.0) for d in ids) / max(len(ids), 1) / 1e6))
PY
