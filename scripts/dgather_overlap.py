"""Profiling experiment: kernel D (strips) and the E + F gather of mr_render_backward one behind the other (as the entry
point runs them) against both at once on two streams (separate outputs): what overlapping them inside the call could win."""
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
lib = _lib.load(); P = _lib.ptr
f32 = dict(dtype=torch.float32, device=dev)
rgb, alpha, depth = torch.empty((B, 3, is_, is_), **f32), torch.empty((B, is_, is_), **f32), torch.empty((B, is_, is_), **f32)
fim = torch.empty((B, is_, is_), dtype=torch.int32, device=dev); wmap = torch.empty((B, is_, is_, 3), **f32)
wbytes = int(lib.mr_render_workspace_bytes(B, F, is_)); work = torch.empty((wbytes,), dtype=torch.uint8, device=dev)
bg = torch.zeros(3, **f32)
st0 = _lib.stream_ptr(dev)
_lib.call("mr_render_forward", P(faces), P(tex2), P(bg), 0, P(rgb), P(alpha), P(depth), P(fim), P(wmap), None, P(work), wbytes, B, F, is_, 2, 0.1, 100.0, 1e-3, 1, 1, 1, 0, st0)
g_rgb, g_alpha, g_depth = torch.randn_like(rgb), torch.randn_like(alpha), torch.randn_like(depth)
gfA, gfB, gtex = torch.empty_like(faces), torch.empty_like(faces), torch.empty_like(tex2)
bw = int(lib.mr_render_backward_workspace_bytes(B, F, is_))
wA, wB = torch.empty((bw,), dtype=torch.uint8, device=dev), torch.empty((bw,), dtype=torch.uint8, device=dev)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()

def d_only(stream):
    _lib.call("mr_render_backward", P(faces), None, P(fim), P(rgb), P(alpha), P(g_rgb), P(g_alpha), None, P(gfA), None, P(wA), bw,
              B, F, is_, 2, 0.1, 100.0, 1e-3, 1, 1, 0, 0, stream.cuda_stream)

def ef_only(stream):
    _lib.call("mr_render_backward", P(faces), P(tex2), P(fim), None, None, P(g_rgb), None, P(g_depth), P(gfB), P(gtex), P(wB), bw,
              B, F, is_, 2, 0.1, 100.0, 1e-3, 1, 0, 1, 0, stream.cuda_stream)

def timed(fn, n=20):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a.record()
    for _ in range(n): fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

cur = torch.cuda.current_stream()
def sequential():
    d_only(cur); ef_only(cur)
def concurrent():
    sA.wait_stream(cur); sB.wait_stream(cur)
    d_only(sA); ef_only(sB)
    cur.wait_stream(sA); cur.wait_stream(sB)
print("D alone %.1f us, E+F alone %.1f us" % (timed(lambda: d_only(cur)), timed(lambda: ef_only(cur))))
print("one behind the other %.1f us, on two streams %.1f us" % (timed(sequential), timed(concurrent)))
