"""Workgroup timeline of the flow-mode forward tile kernel at the training shape (2B = 128 meshes, 256x256): when every
tile with geometry started and ended, on which compute unit, with how many candidate faces.  Needs a profiling build:
    HOC_HIPCC_FLAGS=-DMR_WG_TIMELINE python -c "import __graft_entry__ as g; g.build()" && python scripts/wg_timeline.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from handobjectconsist_amd import _lib
from handobjectconsist_amd.neurender import nr_ops
from handobjectconsist_amd.utils import synth
dev = torch.device("cuda:0")
B, is_ = int(os.environ.get("HOC_TL_BATCH", "64")), int(os.environ.get("HOC_TL_SIZE", "256"))
s = synth.random_scene(B, seed=0, image_size=is_)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
eye, z3, z5 = torch.eye(3, device=dev)[None], torch.zeros(1, 3, device=dev), torch.zeros(1, 5, device=dev)
v1 = nr_ops.projection(t(s["verts1"]), t(s["K1"]), eye, z3, z5, is_)
v2 = nr_ops.projection(t(s["verts2"]), t(s["K2"]), eye, z3, z5, is_)
fidx = t(s["faces"]).to(torch.int32)
pv, pf = torch.cat([v1, v2], 0).contiguous(), torch.cat([fidx, fidx], 0).contiguous()
B2, V, F0 = pv.shape[0], pv.shape[1], pf.shape[1]
cols = torch.randn(B2, V, 3, device=dev)
lib = _lib.load(); st = _lib.stream_ptr(dev); P = _lib.ptr
f32 = dict(dtype=torch.float32, device=dev)
rgb, alpha, depth, mask = torch.empty((B2, 3, is_, is_), **f32), torch.empty((B2, is_, is_), **f32), torch.empty((B2, is_, is_), **f32), torch.empty((B2, is_, is_), **f32)
fim, wmap = torch.empty((B2, is_, is_), dtype=torch.int32, device=dev), torch.empty((B2, is_, is_, 3), **f32)
hit = torch.empty((B2, is_ // 8, is_ // 32, 4), dtype=torch.uint8, device=dev)
wbytes = int(lib.mr_render_workspace_bytes(B2, 2 * F0, is_)); work = torch.empty((wbytes,), dtype=torch.uint8, device=dev)
bg = torch.zeros(3, **f32)
lut = torch.ones(2 * F0 + 2, **f32)
flush = torch.zeros(768 * 1024 * 1024 // 4, **f32)
FUSED = os.environ.get("HOC_FUSED_RECORDS", "1") == "1"  # as the training step launches it: per-face pass inside the binning kernel
hdr_off = _lib.tile_list(work, B2, 2 * F0, is_)[0].value - work.data_ptr()
clear = work[hdr_off:hdr_off + int(lib.mr_render_clear_bytes(B2, 2 * F0, is_))]
XFLAGS = int(os.environ.get("HOC_FWD_FLAGS", "0"))
def fn():
    if FUSED:
        clear.zero_()
    _lib.call("mr_render_flow_forward", P(pv), P(pf), P(cols), P(bg), 0, P(lut), int(lut.numel()), 0.99999, P(rgb), P(alpha), P(mask), P(depth), P(wmap), P(fim), P(hit), P(work), wbytes, B2, V, F0, 1, is_, 0.1, 100.0, 1e-3, _lib.FLAG_SPARSE_TILES | (_lib.FLAG_TILE_LIST_CLEARED if FUSED else 0) | XFLAGS, None, -1, None, None, 0, 0, st)
print("cold %.1f us" % (bench.event_time_ms(fn, 10, flush=flush) * 1e3))
flush.add_(1.0); torch.cuda.synchronize()
fn(); torch.cuda.synchronize()
# phases of the binning pass (one workgroup per image): mean over the images, microseconds
bb = np.zeros(1024 * 8, dtype=np.uint64)
lib.mr_debug_bin_times.argtypes = [ctypes.c_void_p, ctypes.c_long]
assert lib.mr_debug_bin_times(bb.ctypes.data, bb.nbytes) == 0
bt_all = bb.reshape(1024, 8).astype(np.int64)
bt_all = bt_all[bt_all[:, 5] > 0]  # workgroups of the last launch (parts per image since round 5; surplus ones leave at once)
exch = bool((bt_all[:, 6] > 0).any())
# stamps in time order: 0 start, 1 counters zeroed, 2 (per-face pass +) counting pass done, [6 counters published, 7 barrier
# passed,] 3 scan + headers done, 4 tile list written, 5 fill pass done
order = [0, 1, 2, 6, 7, 3, 4, 5] if exch else [0, 1, 2, 3, 4, 5]
names = (["zero counters (+ zero_fill)", "(per-face pass +) count", "publish counters", "barrier", "read parts + scan + bin headers",
          "tile list (part 0)", "pass 2: fill"] if exch else
         ["zero counters (+ zero_fill)", "(per-face pass +) count", "scan + bin headers", "tile list", "pass 2: fill"])
bt = bt_all[:, order]
print("binning pass: %d workgroups; per workgroup (us):" % len(bt), {n_: round(float((bt[:, k + 1] - bt[:, k]).mean()) * 0.01, 2) for k, n_ in enumerate(names)},
      "total %.2f" % (float((bt[:, -1] - bt[:, 0]).mean()) * 0.01), "first start -> last end %.2f" % ((bt[:, -1].max() - bt[:, 0].min()) * 0.01))
n = 32768
buf = np.zeros(n * 4, dtype=np.uint64)
lib.mr_debug_times.argtypes = [ctypes.c_void_p, ctypes.c_long]
assert lib.mr_debug_times(buf.ctypes.data, buf.nbytes) == 0
a = buf.reshape(n, 4)
g = a[a[:, 0] != 0]
t0 = g[:, 0].astype(np.int64); t1 = g[:, 1].astype(np.int64)
nrec = (g[:, 2] >> np.uint64(32)).astype(np.int64); hw = (g[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
base = t0.min()
tick_us = 0.01  # 100 MHz
s_ = (t0 - base) * tick_us; e_ = (t1 - base) * tick_us; d = e_ - s_
print("geometry WGs", len(g), "first start 0, last start %.1f us, last end %.1f us" % (s_.max(), e_.max()))
print("duration us: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (d.mean(), np.percentile(d, 50), np.percentile(d, 90), np.percentile(d, 99), d.max()))
print("n_rec: mean %.0f p50 %.0f p90 %.0f max %d ; corr(duration, n_rec) %.2f" % (nrec.mean(), np.percentile(nrec, 50), np.percentile(nrec, 90), nrec.max(), np.corrcoef(d, nrec)[0, 1]))
for lo, hi_ in ((0, 50), (50, 100), (100, 200), (200, 400), (400, 2000)):
    m = (nrec >= lo) & (nrec < hi_)
    if m.any(): print("  n_rec [%d,%d): %d WGs, duration mean %.1f us" % (lo, hi_, m.sum(), d[m].mean()))
edges = np.arange(0, e_.max() + 5, 5.0)
print("active geometry WGs at t (us):")
for x in edges:
    print("  t=%5.0f  active %5d  started so far %5d" % (x, int(((s_ <= x) & (e_ > x)).sum()), int((s_ <= x).sum())))
cu = ((hw >> 24) & 0xff) * 1000 + ((hw >> 13) & 0x7) * 100 + ((hw >> 8) & 0xf)  # xcc, se_id, cu_id
u, cnt = np.unique(cu, return_counts=True)
busy = np.array([d[cu == k].sum() for k in u])
print("distinct (xcc,se,cu) %d ; geometry WGs per CU: min %d mean %.1f max %d ; sum of durations per CU us: min %.0f mean %.0f max %.0f" % (len(u), cnt.min(), cnt.mean(), cnt.max(), busy.min(), busy.mean(), busy.max()))
late = np.argsort(-e_)[:10]
print("last finishers: ", [(round(float(s_[i]), 1), round(float(e_[i]), 1), int(nrec[i])) for i in late])
wg = g[:, 3].astype(np.int64)
gaps = []
for k in np.unique(wg):
    m = np.where(wg == k)[0]
    o = m[np.argsort(s_[m])]
    gaps += list(s_[o][1:] - e_[o][:-1])
gaps = np.array(gaps)
if len(gaps):
    print("between the end of a tile's scan + drain and the start of the workgroup's next tile (resolve + fetch): mean %.1f p50 %.1f p90 %.1f max %.1f us over %d gaps" % (gaps.mean(), np.percentile(gaps, 50), np.percentile(gaps, 90), gaps.max(), len(gaps)))
