"""Phase stamps of kernel D by strips (pixel_map_strip_kernel) on a -DMR_WG_TIMELINE build: per workgroup start / strip
staged / owners listed / done, with the counts of queued entries, sweeps, 64-record sweep steps and header batches.
    HOC_HIPCC_FLAGS=-DMR_WG_TIMELINE python handobjectconsist_amd/build.py && HOC_HIPCC_FLAGS=-DMR_WG_TIMELINE python scripts/dstrip_timeline.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from handobjectconsist_amd import _lib

dev = torch.device("cuda:0")
out = bench.kernel_bench(dev, 64, 256, 5, ("render_backward_full(D+E+F)",))
print({k: (v["ms"], v["ms_cache_warm"]) for k, v in out.items()})
lib = _lib.load()
buf = np.zeros(8192 * 16, dtype=np.uint64)
lib.mr_debug_ps_times.argtypes = [ctypes.c_void_p, ctypes.c_long]
assert lib.mr_debug_ps_times(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(8192, 16).astype(np.int64)
t = t[t[:, 3] > 0]
print('workgroups that passed the range test', len(t), 'with entries', int((t[:, 4] > 0).sum()))
t0 = t[:, 0].min()
us = lambda a: np.round(a * 0.01, 2)
print("workgroups", len(t))
for k, n in enumerate(["stage", "list owners", "process"]):
    d = (t[:, k + 1] - t[:, k]) * 0.01
    print(f"  {n:12s} mean {d.mean():7.2f} us  median {np.median(d):7.2f}  p90 {np.percentile(d, 90):7.2f}  max {d.max():7.2f}")
tot = (t[:, 3] - t[:, 0]) * 0.01
print(f"  total        mean {tot.mean():7.2f} us  median {np.median(tot):7.2f}  p90 {np.percentile(tot, 90):7.2f}  max {tot.max():7.2f}")
for k, n in enumerate(["entries", "chunk tasks", "terms", "batches"]):
    c = t[:, 4 + k]
    print(f"  {n:32s} mean {c.mean():8.1f}  max {c.max():6d}  total {c.sum()}")
print("launch: last start %.1f us, last end %.1f us" % ((t[:, 0].max() - t0) * 0.01, (t[:, 3].max() - t0) * 0.01))
for x in range(0, 500, 40):
    s_, e_ = (t[:, 0] - t0) * 0.01, (t[:, 3] - t0) * 0.01
    print("  t=%3d us  resident %4d  started %4d" % (x, int(((s_ <= x) & (e_ > x)).sum()), int((s_ <= x).sum())))
