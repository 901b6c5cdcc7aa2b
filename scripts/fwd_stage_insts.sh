#!/bin/bash
# per-stage instruction counts of raster_tile_kernel (see scripts/fwd_stage_insts.py); output: gpurun_out/fwd_stage_insts.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/stage_pmc
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d /tmp/stage_pmc -o p -- python $ROOT/scripts/fwd_stage_insts.py > $ROOT/gpurun_out/fwd_stage_pmc.log 2>&1
f=$(find /tmp/stage_pmc -name "p_counter_collection.csv" | head -1)
python - "$f" $ROOT <<'PY' | tee $ROOT/gpurun_out/fwd_stage_insts.txt
import csv, sys
sys.path.insert(0, sys.argv[2] + "/scripts")
from collections import defaultdict, OrderedDict
rows = defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    if "raster_tile_kernel<true, true>" in r["Kernel_Name"]:
        rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
import importlib.util
spec = importlib.util.spec_from_file_location("v", sys.argv[2] + "/scripts/fwd_stage_insts.py"); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
names = [n for n, _ in m.VARIANTS]
assert len(ids) == 3 * len(names), (len(ids), len(names))
cols = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES"]
print("%-44s " % "variant (millions per launch)" + " ".join("%12s" % c[3:] for c in cols))
for i, n in enumerate(names):
    grp = [rows[d] for d in ids[3 * i + 1:3 * i + 3]]
    print("%-44s " % n + " ".join("%12.2f" % (sum(g.get(c, 0.0) for g in grp) / len(grp) / 1e6) for c in cols))
PY
