"""The full raster backward (D + E + F) at the metric shape alone (profiling aid for scripts/pmc_one.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
d = bench.kernel_bench(torch.device("cuda:0"), 64, 256, 5, only=("render_backward_full(D+E+F)",))
for k, v in d.items():
    print('%-58s %7.1f %7.1f' % (k, v['ms'] * 1000, v['ms_cache_warm'] * 1000))
