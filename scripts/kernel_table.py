"""python bench.py --kernels-only | python scripts/kernel_table.py"""
import json, sys
txt = sys.stdin.read()
d = json.loads(txt[txt.index("{"):])
for k, v in d.items():
    print(f"{k:32s} cold {v['ms'] * 1e3:8.1f} us  warm {v['ms_cache_warm'] * 1e3:8.1f} us  {v['GBps']:8.1f} GB/s  frac {v['frac_hbm_peak']:.3f}")
