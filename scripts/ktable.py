"""Print bench.py --kernels-only JSON as a table:  python scripts/ktable.py gpurun_out/k1.json [baseline.json]"""
import json
import sys

d = json.load(open(sys.argv[1]))
base = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else {}
if "kernels" in d:
    d = d["kernels"]
if "kernels" in base:
    base = base["kernels"]
for k, v in d.items():
    b = base.get(k)
    extra = f"   (was {b['ms'] * 1000:7.1f} / {b['ms_cache_warm'] * 1000:7.1f})" if b else ""
    print(f"{k:52s} {v['ms'] * 1000:8.1f} us cold {v['ms_cache_warm'] * 1000:8.1f} us warm  frac {v['frac_hbm_peak']:.3f}{extra}")
