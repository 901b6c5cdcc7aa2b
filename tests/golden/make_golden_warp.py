"""Generate tests/golden/warp_*.npz by importing the REAL reference on CPU.

Runs only in the build container (needs /root/reference); the committed .npz files are
data (seeded inputs + the reference's outputs), nothing of the reference's source.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_warp.py

Imports (all execute on CPU with torch 2.10): meshreg.warping.imgflowarp (all five
functions), meshreg.optim.lossutils, meshreg.optim.pyramidloss (kornia stubbed: its
symbols are only touched by the ssim / level_nb>1 branches), meshreg.models.project.
"""
import os
import sys
import types
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
k, kl = types.ModuleType("kornia"), types.ModuleType("kornia.losses")
kg, kt = types.ModuleType("kornia.geometry"), types.ModuleType("kornia.geometry.transform")
kl.SSIM = object
kt.ScalePyramid = lambda: None
k.losses, k.geometry, kg.transform = kl, kg, kt
sys.modules.update({"kornia": k, "kornia.losses": kl, "kornia.geometry": kg,
                    "kornia.geometry.transform": kt})
torch.Tensor.cuda = lambda self, *a, **kw: self  # imgflowarp.py:80-85 call .cuda() unconditionally

from meshreg.warping import imgflowarp  # noqa: E402
from meshreg.optim import pyramidloss, lossutils  # noqa: E402
from meshreg.models import project  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(4)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def make_flows(rng, B, H, W, amp=3.0, zero_frac=0.3):
    """[B,H,W,2] flows: smooth-ish sub-pixel motion, some exactly-zero pixels (invalid), some
    exactly-integer offsets, some far out-of-bounds vectors."""
    f = (rng.standard_normal((B, H, W, 2)) * amp).astype(np.float32)
    f[rng.random((B, H, W)) < zero_frac] = 0
    integer = rng.random((B, H, W)) < 0.1
    f[integer] = np.round(f[integer])
    far = rng.random((B, H, W)) < 0.03
    f[far] += 1000.0
    xonly = rng.random((B, H, W)) < 0.05  # dx == 0, dy != 0
    f[xonly, 0] = 0
    return f


def jitter(rng, B, H, W, C=3):
    m = np.ones((B, C, H, W), np.float32)
    for b in range(B):
        l, r, u, d = rng.integers(0, 4, size=4)
        if l: m[b, :, :, :l] = 0
        if r: m[b, :, :, -r:] = 0
        if u: m[b, :, :u, :] = 0
        if d: m[b, :, -d:, :] = 0
    return m


def gen_warp():
    rng = np.random.default_rng(0)
    B, C, H, W = 2, 3, 17, 23
    x = rng.uniform(-0.5, 0.5, (B, C, H, W)).astype(np.float32)
    flow = make_flows(rng, B, H, W).transpose(0, 3, 1, 2).copy()
    gout = rng.standard_normal((B, C, H, W)).astype(np.float32)
    res = {"x": x, "flow": flow, "grad_out": gout}
    for mode in ("bilinear", "nearest"):
        xt, ft = t(x).requires_grad_(True), t(flow).requires_grad_(True)
        out, mask = imgflowarp.warp(xt, ft, mode=mode)
        res[f"out_{mode}"] = out.detach().numpy()
        res[f"mask_{mode}"] = mask.detach().numpy()
        if mode == "bilinear":
            (out * t(gout)).sum().backward()
            res["grad_x"] = xt.grad.numpy()
            res["grad_flow"] = ft.grad.numpy()
    for scale in (False, True):
        res[f"meshgrid_{int(scale)}"] = imgflowarp.get_spatial_meshgrid(t(x), scale=scale).numpy()
    np.savez_compressed(os.path.join(OUT, "warp_basic.npz"), **res)


def gen_occlusion():
    rng = np.random.default_rng(1)
    B, H, W = 2, 20, 28
    # consistent flows inside a blob, inconsistent elsewhere
    m1 = np.zeros((B, 1, H, W), np.float32)
    m2 = np.zeros((B, 1, H, W), np.float32)
    m1[:, :, 4:15, 5:20] = 1
    m2[:, :, 5:16, 7:22] = 1
    m2[1] *= rng.random((1, H, W)).astype(np.float32)  # raw-alpha style non-binary mask (Q4)
    f12 = np.zeros((B, 3, H, W), np.float32)
    f21 = np.zeros((B, 3, H, W), np.float32)
    f12[:, 0], f12[:, 1], f12[:, 2] = 2.0, 1.0, 1.0
    f21[:, 0], f21[:, 1], f21[:, 2] = -2.0, -1.0, 1.0
    f12[:, :2] += (rng.standard_normal((B, 2, H, W)) * 0.2).astype(np.float32)
    f21[1, :2] += (rng.standard_normal((2, H, W)) * 2.0).astype(np.float32)  # inconsistent sample
    f12 *= m1
    f21 *= (m2 > 0)
    o1, o2 = imgflowarp.get_occlusion_mask(t(m1), t(m2), t(f12), t(f21))
    np.savez_compressed(os.path.join(OUT, "warp_occlusion.npz"), mask_flow1=m1, mask_flow2=m2,
                        flow12=f12, flow21=f21, occl1=o1.numpy(), occl2=o2.numpy())


def gen_pair_consist():
    rng = np.random.default_rng(2)
    B, H, W = 3, 19, 26
    crit = pyramidloss.PyramidCriterion("l1")
    image_ref = rng.uniform(-0.5, 0.5, (B, 3, H, W)).astype(np.float32)
    image = rng.uniform(-0.5, 0.5, (B, 3, H, W)).astype(np.float32)
    jm_ref, jm = jitter(rng, B, H, W), jitter(rng, B, H, W)
    f12 = make_flows(rng, B, H, W, amp=1.5)
    f21 = make_flows(rng, B, H, W, amp=1.5)
    f12[2] = 0  # a sample with no valid pixel in the backward direction
    f21[2] = 0  # ... nor forward: masked mean divides by 1
    gl = rng.uniform(0.5, 1.5, (B,)).astype(np.float32)
    res = dict(image_ref=image_ref, image=image, jitter_ref=jm_ref, jitter=jm, flow12=f12, flow21=f21,
               grad_loss=gl)
    for ub in (False, True):
        a, b = t(f12).requires_grad_(True), t(f21).requires_grad_(True)
        loss, masks, warps, diffs = imgflowarp.pair_consist(
            [a, b], t(image_ref), t(image), t(jm_ref), t(jm), crit, use_backward=ub)
        (loss * t(gl)).sum().backward()
        tag = f"ub{int(ub)}"
        res[f"loss_{tag}"] = loss.detach().numpy()
        res[f"grad_flow12_{tag}"] = a.grad.numpy() if a.grad is not None else np.zeros_like(f12)
        res[f"grad_flow21_{tag}"] = b.grad.numpy() if b.grad is not None else np.zeros_like(f21)
        if ub:
            for i in (0, 1):
                res[f"warp_mask{i + 1}"] = masks[i]["warp_mask"].detach().numpy()
                res[f"full_mask{i + 1}"] = masks[i]["full_mask"].detach().numpy()
                res[f"flow_mask{i + 1}"] = masks[i]["flow_mask"].detach().numpy()
                res[f"warp{i + 1}"] = warps[i].detach().numpy()
                res[f"diff{i + 1}"] = diffs[i].detach().numpy()
    np.savez_compressed(os.path.join(OUT, "warp_pair_consist.npz"), **res)


def gen_misc():
    rng = np.random.default_rng(3)
    d = rng.random((4, 3, 9, 11)).astype(np.float32)
    m = (rng.random((4, 3, 9, 11)) < 0.4)
    m[1] = False
    res = dict(dists=d, mask=m, masked_mean=lossutils.batch_masked_mean_loss(t(d), t(m)).numpy())
    pts = rng.standard_normal((4, 7, 3)).astype(np.float32) * 0.05
    K = np.tile(np.array([[350.0, 0, 130.0], [0, 352.0, 125.0], [0, 0, 1]], np.float32), (4, 1, 1))
    sc = rng.uniform(-1e-4, 1e-4, (4, 1)).astype(np.float32)
    tr = rng.uniform(-20, 20, (4, 2)).astype(np.float32)
    r3d, c3d = project.recover_3d_proj(t(pts), t(K), t(sc), t(tr), off_z=0.4, input_res=(256, 256))
    res.update(objpoints3d=pts, camintr=K, est_scale=sc, est_trans=tr, recons3d=r3d.numpy(),
               est_c3d=c3d.numpy())
    np.savez_compressed(os.path.join(OUT, "warp_misc.npz"), **res)


if __name__ == "__main__":
    gen_warp()
    gen_occlusion()
    gen_pair_consist()
    gen_misc()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
