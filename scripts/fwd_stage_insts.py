"""Instruction counts of the forward tile kernel by stage: the training-shape flow-mode launch (2B = 128 meshes, 256x256)
with stages disabled through the kernel's dbg bits, each variant launched 3 times.  Run under
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace ...
and read with scripts/fwd_stage_insts.sh (dispatch order = variant order)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from handobjectconsist_amd import _lib
from handobjectconsist_amd.neurender import nr_ops
from handobjectconsist_amd.utils import synth

VARIANTS = (("full", 0), ("no resolve", 2), ("S1 only (no S2, S3, resolve)", 2 + 8), ("nothing per record (no S1)", 2 + 4),
            ("S1 + S2 without span search / emission", 2 + 16 + 32 + 64), ("S1 + S2 search, no emission", 2 + 16 + 64),
            ("S1 + S2 (no S3)", 2 + 16), ("S1 + S2 + S3 (= no resolve)", 2))
if __name__ == "__main__":
    dev = torch.device("cuda:0")
    B, is_ = 64, 256
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
    rgb, alpha, depth, mask = (torch.empty((B2, 3, is_, is_), **f32), torch.empty((B2, is_, is_), **f32),
                               torch.empty((B2, is_, is_), **f32), torch.empty((B2, is_, is_), **f32))
    fim, wmap = torch.empty((B2, is_, is_), dtype=torch.int32, device=dev), torch.empty((B2, is_, is_, 3), **f32)
    hit = torch.empty((B2, is_ // 8, is_ // 32, 4), dtype=torch.uint8, device=dev)
    wbytes = int(lib.mr_render_workspace_bytes(B2, 2 * F0, is_)); work = torch.empty((wbytes,), dtype=torch.uint8, device=dev)
    bg, lut = torch.zeros(3, **f32), torch.ones(2 * F0 + 2, **f32)
    for name, dbg in VARIANTS:
        for _ in range(3):
            _lib.call("mr_render_flow_forward", P(pv), P(pf), P(cols), P(bg), 0, P(lut), int(lut.numel()), 0.99999, P(rgb),
                      P(alpha), P(mask), P(depth), P(wmap), P(fim), P(hit), P(work), wbytes, B2, V, F0, 1, is_, 0.1, 100.0,
                      1e-3, _lib.FLAG_SPARSE_TILES | (dbg << 8), None, -1, None, None, 0, 0, st)  # listed launch, a quarter of the tiles as the guess
        torch.cuda.synchronize()
