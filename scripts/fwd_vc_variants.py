"""Profiling experiment: the training launch of the vertex-colour forward (2B = 128 meshes, 256x256) with parts
of the pipeline disabled (dbg bits of raster_tile_kernel; bit 128 = binning pass alone)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from handobjectconsist_amd import _lib
from handobjectconsist_amd.neurender import nr_ops
from handobjectconsist_amd.utils import synth

dev = torch.device("cuda:0")
B, is_ = int(os.environ.get("B", 64)), int(os.environ.get("IS", 256))
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
rgb, alpha, depth = torch.empty((B2, 3, is_, is_), **f32), torch.empty((B2, is_, is_), **f32), torch.empty((B2, is_, is_), **f32)
fim, wmap = torch.empty((B2, is_, is_), dtype=torch.int32, device=dev), torch.empty((B2, is_, is_, 3), **f32)
wbytes = int(lib.mr_render_workspace_bytes(B2, 2 * F0, is_)); work = torch.empty((wbytes,), dtype=torch.uint8, device=dev)
bg = torch.zeros(3, **f32)
flush = torch.zeros(768 * 1024 * 1024 // 4, **f32)
VARIANTS = (("full", 0), ("binning pass alone", 128), ("setup: no RecVerts stores", 128 + (16 << 16)), ("setup: no inverse", 128 + (32 << 16)), ("setup: neither", 128 + (48 << 16)),
                  ("setup: no bin fill pass", 128 + (2 << 16)), ("setup: records only, no stores, no fill", 128 + (50 << 16)),
                  ("no scan (background fill only)", 1), ("geometry tiles only (no background stores)", 256),
                  ("geometry tiles only, no resolve", 256 + 2), ("scan+drain, no resolve", 2),
                  ("scan only (no drain), no resolve", 6), ("nothing (zbuf init only)", 3), ("scan + S1, no resolve", 10),
                  ("scan + S1 + S2 (no shade), no resolve", 18), ("... S2 without search (bbox spans)", 18 + 32),
                  ("... S2 without emission", 18 + 64), ("... S2 without search and emission", 18 + 96))
if os.environ.get("ONLY_BIN"):
    VARIANTS = (("binning pass alone", 128),)
if os.environ.get("ONLY_FULL"):
    VARIANTS = (("full", 0),)
for name, dbg in VARIANTS:
    fn = lambda: _lib.call("mr_render_vc_forward", P(pv), P(pf), P(cols), P(bg), 0, P(rgb), P(alpha), P(depth), P(fim),
                           P(wmap), P(work), wbytes, B2, V, F0, 1, is_, 0.1, 100.0, 1e-3, 1, 1, 1, dbg << 8, 0, st)
    print(f"{name:42s} {bench.event_time_ms(fn, 20, flush=flush) * 1e3:8.1f} us cold {bench.event_time_ms(fn, 30) * 1e3:8.1f} us warm")
