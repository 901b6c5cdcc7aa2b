#!/bin/bash
# One quick GPU call of round 5:  scripts/r5_quick.sh <tag> "<pytest selection or empty>" ["<kernel groups ; separated>"]
# -> gpurun_out/<tag>/{pytest.txt, instep_256.json, instep_480.json, kernels_*.json}
TAG=${1:-q}
SEL=$2
GROUPS_=$3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
if [ -n "$SEL" ]; then timeout 1500 python -m pytest $SEL -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest.txt; tail -3 $OUT/pytest.txt; fi
timeout 600 python scripts/instep.py > $OUT/instep_256.json 2> $OUT/instep_256.err
timeout 600 python scripts/instep.py --batch 8 --image-size 480 --image-height 270 > $OUT/instep_480.json 2> $OUT/instep_480.err
if [ -n "$GROUPS_" ]; then
  HOC_KERNEL_GROUPS="$GROUPS_" timeout 600 python bench.py --kernels-only > $OUT/kernels_256.json 2> $OUT/kernels_256.err
  HOC_KERNEL_GROUPS="$GROUPS_" timeout 600 python bench.py --kernels-only --batch 8 --image-size 480 > $OUT/kernels_480.json 2> $OUT/kernels_480.err
  HOC_KERNEL_GROUPS="$GROUPS_" timeout 600 python bench.py --kernels-only --batch 32 --image-size 640 > $OUT/kernels_640.json 2> $OUT/kernels_640.err
fi
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/instep_*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], {k: v["median_us"] for k, v in d.items()})
    except Exception as e:
        print(f, "unreadable", e)
for f in sorted(glob.glob("$OUT/kernels_*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], {k[:40]: (v.get("ms"), v.get("ms_cache_warm")) for k, v in d.items() if isinstance(v, dict)})
    except Exception as e:
        print(f, "unreadable", e)
PY
