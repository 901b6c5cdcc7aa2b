"""Profiling experiment: kernel D (mr_render_backward with grad_faces only) with parts disabled."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from handobjectconsist_amd import _lib
from handobjectconsist_amd.neurender import nr_ops
from handobjectconsist_amd.utils import synth, textutils

dev = torch.device("cuda:0")
B, is_ = 64, 256
s = synth.random_scene(B, seed=0, image_size=is_)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
verts, faces_idx, K = t(s["verts1"]), t(s["faces"]), t(s["K1"])
colors = torch.randn(B, verts.shape[1], 3, device=dev)
v = nr_ops.projection(verts, K, torch.eye(3, device=dev)[None], torch.zeros(1, 3, device=dev), torch.zeros(1, 5, device=dev), is_)
f2 = torch.cat((faces_idx, faces_idx.flip(-1)), 1)
faces = nr_ops.vertices_to_faces(v, f2).contiguous()
tex = textutils.batch_vertex_textures(faces_idx, colors)
tex2 = torch.cat((tex, tex.permute(0, 1, 4, 3, 2, 5)), 1).contiguous()
F = faces.shape[1]
lib = _lib.load(); st = _lib.stream_ptr(dev); P = _lib.ptr
f32 = dict(dtype=torch.float32, device=dev)
rgb, alpha, depth = torch.empty((B, 3, is_, is_), **f32), torch.empty((B, is_, is_), **f32), torch.empty((B, is_, is_), **f32)
fim = torch.empty((B, is_, is_), dtype=torch.int32, device=dev); wmap = torch.empty((B, is_, is_, 3), **f32)
wbytes = int(lib.mr_render_workspace_bytes(B, F, is_)); work = torch.empty((wbytes,), dtype=torch.uint8, device=dev)
bg = torch.zeros(3, **f32)
_lib.call("mr_render_forward", P(faces), P(tex2), P(bg), 0, P(rgb), P(alpha), P(depth), P(fim), P(wmap), None, P(work), wbytes, B, F, is_, 2, 0.1, 100.0, 1e-3, 1, 1, 1, 0, st)
g_rgb, g_alpha = torch.randn_like(rgb), torch.randn_like(alpha)
grad_faces = torch.empty_like(faces)
bw = int(lib.mr_render_backward_workspace_bytes(B, F, is_)); bwork = torch.empty((bw,), dtype=torch.uint8, device=dev)
flush = torch.zeros(768 * 1024 * 1024 // 4, **f32)
for name, dbg, ws in (("strips: full", 0, True), ("strips: no chunk tasks", 1, True), ("strips: no lane in-sweeps", 2, True), ("strips: headers only", 3, True),
                      ("strips: listing + staging only", 4, True), ("strips: tasks without their steps", 8, True), ("planes kernel", 0, False)):
    fn = lambda: _lib.call("mr_render_backward", P(faces), None, P(fim), P(rgb), P(alpha), P(g_rgb), P(g_alpha), None, P(grad_faces), None,
                           P(bwork) if ws else None, bw if ws else 0, B, F, is_, 2, 0.1, 100.0, 1e-3, 1, 1, 0, dbg << 8, st)
    print(f"{name:34s} cold {bench.event_time_ms(fn, 10, flush=flush) * 1e3:8.1f} us   warm {bench.event_time_ms(fn, 10) * 1e3:8.1f} us")
cov = fim >= 0
print("covered px", int(cov.sum()), "distinct faces", int(sum(len(torch.unique(fim[i][cov[i]])) for i in range(B))))
