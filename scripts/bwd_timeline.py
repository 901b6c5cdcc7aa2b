"""Phase stamps of the training-path raster backward (scatter_tiles_kernel) on a -DMR_WG_TIMELINE build: per workgroup
start / covered-tile list built / table zeroed + maximum pass / main pass / flushed, and the launch's shape in time.
    HOC_HIPCC_FLAGS=-DMR_WG_TIMELINE python handobjectconsist_amd/build.py && HOC_HIPCC_FLAGS=-DMR_WG_TIMELINE python scripts/bwd_timeline.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from handobjectconsist_amd import _lib

dev = torch.device("cuda:0")
out = bench.kernel_bench(dev, 64, 256, 10, (bench.ROOF_BWD, bench.ROOF_FWD))
print({k: (v["ms"], v["ms_cache_warm"]) for k, v in out.items()})
lib = _lib.load()
buf = np.zeros(4096 * 8, dtype=np.uint64)
lib.mr_debug_st_times.argtypes = [ctypes.c_void_p, ctypes.c_long]
assert lib.mr_debug_st_times(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(4096, 8)[:, :5].astype(np.int64)
t = t[t[:, 0] > 0]  # the workgroups of the last launch (8 per image)
t0 = t[:, 0].min()
worked = t[:, 4] > 0
print("workgroups", len(t), "that walked tiles", int(worked.sum()))
names = ["covered-tile list", "zero table + maximum pass", "main pass", "flush"]
w = t[worked]
print("per working workgroup (us):", {n: round(float((w[:, k + 1] - w[:, k]).mean()) * 0.01, 2) for k, n in enumerate(names)},
      "total %.2f" % (float((w[:, 4] - w[:, 0]).mean()) * 0.01))
for k, n in enumerate(names):
    d = (w[:, k + 1] - w[:, k]) * 0.01
    print("  %-28s p10 %.2f  p50 %.2f  p90 %.2f  p99 %.2f  max %.2f" % (n, *np.percentile(d, [10, 50, 90, 99]), d.max()))
e = (w[:, 4] - t0) * 0.01
print("  end of a working workgroup   p10 %.2f  p50 %.2f  p90 %.2f  p99 %.2f  max %.2f" % (*np.percentile(e, [10, 50, 90, 99]), e.max()))
st = (w[:, 0] - t0) * 0.01
print("  start of a working workgroup p10 %.2f  p50 %.2f  p90 %.2f  p99 %.2f  max %.2f" % (*np.percentile(st, [10, 50, 90, 99]), st.max()))
idle = t[~worked]
if len(idle):
    print("idle workgroups: list %.2f us" % (float((idle[:, 1] - idle[:, 0]).mean()) * 0.01))
print("launch: first start 0, last start %.1f us, last end %.1f us" % ((t[:, 0].max() - t0) * 0.01, (np.where(worked, t[:, 4], t[:, 1]).max() - t0) * 0.01))
for x in range(0, 60, 4):
    s_, e_ = (t[:, 0] - t0) * 0.01, (np.where(worked, t[:, 4], t[:, 1]) - t0) * 0.01
    print("  t=%3d us  resident %4d  started %4d" % (x, int(((s_ <= x) & (e_ > x)).sum()), int((s_ <= x).sum())))
