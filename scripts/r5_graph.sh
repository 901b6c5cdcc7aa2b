#!/bin/bash
# graph-replayed step vs eager step: metric config, config 3, config 5 (bf16)   -> gpurun_out/<tag>/
TAG=${1:-g}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
COMMON="--no-kernel-bench --no-cpu-baseline --no-stock-trunk"
timeout 500 python bench.py $COMMON --graph-step > $OUT/graph_256.json 2> $OUT/graph_256.err
timeout 500 python bench.py $COMMON > $OUT/eager_256.json 2> $OUT/eager_256.err
timeout 500 python bench.py $COMMON --batch 8 --image-size 480 --image-height 270 --graph-step > $OUT/graph_480.json 2> $OUT/graph_480.err
timeout 500 python bench.py $COMMON --batch 8 --image-size 480 --image-height 270 > $OUT/eager_480.json 2> $OUT/eager_480.err
timeout 500 python bench.py $COMMON --batch 32 --image-size 640 --image-height 480 --encoder-dtype bf16 --graph-step > $OUT/graph_640.json 2> $OUT/graph_640.err
timeout 500 python bench.py $COMMON --batch 32 --image-size 640 --image-height 480 --encoder-dtype bf16 > $OUT/eager_640.json 2> $OUT/eager_640.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], d.get("step_mode"), d["ms_per_step"], d["value"], d.get("hot_path_ms"))
    except Exception as e:
        print(f, "unreadable", e); print(open(f.replace(".json", ".err")).read()[-1500:])
PY
