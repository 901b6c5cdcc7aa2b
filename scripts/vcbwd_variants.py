"""Profiling experiment: mr_render_vc_backward with parts disabled."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from handobjectconsist_amd import _lib
from handobjectconsist_amd.neurender import nr_ops
from handobjectconsist_amd.utils import synth

dev = torch.device("cuda:0")
B, is_ = 64, 256
s = synth.random_scene(B, seed=0, image_size=is_)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
verts, faces_idx, K = t(s["verts1"]), t(s["faces"]), t(s["K1"])
colors = torch.randn(B, verts.shape[1], 3, device=dev)
v = nr_ops.projection(verts, K, torch.eye(3, device=dev)[None], torch.zeros(1, 3, device=dev), torch.zeros(1, 5, device=dev), is_).contiguous()
fidx = faces_idx.to(torch.int32).contiguous(); F0 = fidx.shape[1]; V = v.shape[1]
lib = _lib.load(); st = _lib.stream_ptr(dev); P = _lib.ptr
f32 = dict(dtype=torch.float32, device=dev)
rgb, alpha, depth = torch.empty((B, 3, is_, is_), **f32), torch.empty((B, is_, is_), **f32), torch.empty((B, is_, is_), **f32)
fim = torch.empty((B, is_, is_), dtype=torch.int32, device=dev); wmap = torch.empty((B, is_, is_, 3), **f32)
wbytes = int(lib.mr_render_workspace_bytes(B, 2 * F0, is_)); work = torch.empty((wbytes,), dtype=torch.uint8, device=dev)
bg = torch.zeros(3, **f32)
_lib.call("mr_render_vc_forward", P(v), P(fidx), P(colors), P(bg), 0, P(rgb), P(alpha), P(depth), P(fim), P(wmap), P(work), wbytes, B, V, F0, 1, is_, 0.1, 100.0, 1e-3, 1, 1, 1, 0, 0, st)
g_rgb = torch.randn_like(rgb); g_cols = torch.empty_like(colors)
flush = torch.zeros(768 * 1024 * 1024 // 4, **f32)
G = 32  # force the face-parallel gather kernel
for name, dbg in (("scatter: full", 0), ("scatter: no flush", 1), ("scatter: no shade", 2), ("scatter: no shade, no flush", 3), ("scatter: no lds atomics", 4), ("scatter: fim pass only", 8), ("scatter: loads + zeroing only", 16), ("scatter, recompute: full", 64),
                  ("gather: full", G), ("gather: no global atomics", G | 1), ("gather: no shade", G | 2),
                  ("gather: no probes (setup only)", G | 7), ("gather: loads only", G | 8), ("gather: loads + boxes", G | 16)):
    fn = lambda: _lib.call("mr_render_vc_backward", P(v), P(fidx), P(fim), P(wmap), P(depth), P(g_rgb), P(g_cols), B, V, F0, 1, is_, 1e-3, dbg << 8, 0, st)
    print(f"{name:30s} cold {bench.event_time_ms(fn, 20, flush=flush) * 1e3:8.1f} us   warm {bench.event_time_ms(fn, 20) * 1e3:8.1f} us")
