"""Large inputs: the tile rasteriser against the brute-force reference-algorithm kernel (which the
randomised sweeps pin to the oracle), where the CPU oracle would take minutes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from handobjectconsist_amd.neurender import rasterize
dev = torch.device("cuda:0")
for B, is_, n, size in ((2, 1024, 60000, 0.01), (1, 2048, 20000, 0.05), (3, 777, 100000, 0.004), (1, 512, 300, 1.5)):
    rng = np.random.default_rng(is_)
    c = rng.uniform(-1.1, 1.1, (B, n, 1, 2)).astype(np.float32)
    faces = np.concatenate([c + rng.uniform(-size, size, (B, n, 3, 2)).astype(np.float32), rng.uniform(0.2, 3, (B, n, 3, 1)).astype(np.float32)], -1)
    faces = np.ascontiguousarray(np.concatenate([faces, faces[:, :, ::-1]], 1))
    tex = rng.uniform(-1, 1, (B, 2 * n, 2, 2, 2, 3)).astype(np.float32)
    f, x = torch.from_numpy(faces).to(dev), torch.from_numpy(tex).to(dev)
    res = {}
    for ref in (False, True):
        rasterize.REFERENCE_ALGO = ref
        torch.cuda.synchronize(); t0 = time.time()
        o = rasterize.rasterize_rgbad(f, x, is_, False, 0.1, 100, 1e-3, (0, 0, 0))
        torch.cuda.synchronize(); res[ref] = (o, time.time() - t0)
    rasterize.REFERENCE_ALGO = False
    a, b = res[False][0], res[True][0]
    same = all(torch.equal(a[k], b[k]) for k in ("face_index_map", "rgb", "alpha", "depth", "weight_map"))
    print(f"B={B} is={is_} faces={2 * n}: identical={same}  covered={float((a['face_index_map'] >= 0).float().mean()):.3f}  tile {res[False][1] * 1e3:.1f} ms  brute {res[True][1] * 1e3:.1f} ms")
