"""Profiling experiment: time mr_render_forward with parts of the tile kernel disabled."""
import sys, os
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
tex = textutils.batch_vertex_textures(faces_idx, torch.randn(B, verts.shape[1], 3, device=dev))
faces_idx2 = torch.cat((faces_idx, faces_idx.flip(-1)), 1)
tex2 = torch.cat((tex, tex.permute(0, 1, 4, 3, 2, 5)), 1).contiguous()
v = nr_ops.projection(verts, K, torch.eye(3, device=dev)[None], torch.zeros(1, 3, device=dev), torch.zeros(1, 5, device=dev), is_)
faces = nr_ops.vertices_to_faces(v, faces_idx2).contiguous()
F = faces.shape[1]
lib = _lib.load(); st = _lib.stream_ptr(dev); P = _lib.ptr
f32 = dict(dtype=torch.float32, device=dev)
rgb, alpha, depth = torch.empty((B, 3, is_, is_), **f32), torch.empty((B, is_, is_), **f32), torch.empty((B, is_, is_), **f32)
fim = torch.empty((B, is_, is_), dtype=torch.int32, device=dev); wmap = torch.empty((B, is_, is_, 3), **f32)
wbytes = int(lib.mr_render_workspace_bytes(B, F, is_)); work = torch.empty((wbytes,), dtype=torch.uint8, device=dev)
bg = torch.zeros(3, **f32)
for name, dbg in (("full", 0), ("no scan (background fill only)", 1), ("scan+drain, no resolve", 2), ("scan only (no drain), no resolve", 6),
                  ("scan only + resolve", 4), ("nothing (zbuf init only)", 3), ("scan + S1, no resolve", 10),
                  ("scan + S1 + S2 (no shade), no resolve", 18), ("... S2 without search (bbox spans)", 18 + 32),
                  ("... S2 without emission", 18 + 64), ("... S2 without search and emission", 18 + 96)):
    fn = lambda: _lib.call("mr_render_forward", P(faces), P(tex2), P(bg), 0, P(rgb), P(alpha), P(depth), P(fim), P(wmap),
                           None, P(work), wbytes, B, F, is_, 2, 0.1, 100.0, 1e-3, 1, 1, 1, dbg << 8, st)
    print(f"{name:40s} {bench.event_time_ms(fn, 30) * 1e3:8.1f} us")
