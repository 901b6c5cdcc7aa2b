"""Phase stamps of the binning launch INSIDE a pair step (bin_boxes_prologue_kernel: vertex stage + per-face pass + counting pass,
exchange, merge + scan + fill by the last arriver) on a -DMR_WG_TIMELINE build, and of the tile kernel's workgroups:
    HOC_LIB_PATH=<timeline build> python scripts/pair_bin_timeline.py [--batch 64] [--image-size 256]
(the per-workgroup means of scripts/wg_timeline.py, taken from the launches scripts/hot_only.py times)."""
import ctypes, os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

sys.argv = [sys.argv[0]] + sys.argv[1:] + ["--no-graph", "--passes", "5"]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "hot_only.py"), run_name="__main__")
import torch
from handobjectconsist_amd import _lib

torch.cuda.synchronize()
lib = _lib.load()
bb = np.zeros(1024 * 8, dtype=np.uint64)
lib.mr_debug_bin_times.argtypes = [ctypes.c_void_p, ctypes.c_long]
assert lib.mr_debug_bin_times(bb.ctypes.data, bb.nbytes) == 0
t = bb.reshape(1024, 8).astype(np.int64)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
print("workgroups", len(t), "of which finished an image", int((t[:, 5] > 0).sum()))
def col(k):
    return (t[:, k] - t0) * 0.01
print("start           p50 %.2f max %.2f" % (np.median(col(0)), col(0).max()))
print("counters zeroed p50 %.2f max %.2f  (+ zero_fill)" % (np.median(col(1)), col(1).max()))
print("count pass done p50 %.2f max %.2f  (vertex stage + per-face pass + counting)" % (np.median(col(2)), col(2).max()))
pub = t[:, 6] > 0
if pub.any():
    print("published       p50 %.2f max %.2f" % (np.median(col(6)[pub]), col(6)[pub].max()))
last = t[:, 5] > 0
for k, n in ((7, "last arriver on"), (3, "scan + headers "), (4, "tile list      "), (5, "fill done      ")):
    m = last & (t[:, k] > 0)
    if m.any():
        print("%s p50 %.2f max %.2f" % (n, np.median(col(k)[m]), col(k)[m].max()))
