"""Parity of the HIP warp kernels (through the C-ABI) against the numpy oracle AND against the
golden vectors captured from the real reference (tests/golden/warp_*.npz)."""
import os

import numpy as np
import pytest
import torch

from handobjectconsist_amd.utils import synth
from oracle import raster_ref as R
from oracle import warp_ref as W

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def close(a, b, rtol, atol, what):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    assert (err <= tol).all(), f"{what}: max err {err.max():.3e}, {(err > tol).sum()} / {err.size} out of tol"


def test_warp_golden(cuda):
    from handobjectconsist_amd.warping import imgflowarp

    g = np.load(os.path.join(GOLDEN, "warp_basic.npz"))
    for mode in ("bilinear", "nearest"):
        x = t(g["x"], cuda).requires_grad_(True)
        flow = t(g["flow"], cuda).requires_grad_(True)
        out, mask = imgflowarp.warp(x, flow, mode=mode)
        assert (mask.cpu().numpy() != g[f"mask_{mode}"]).sum() == 0
        close(out.detach().cpu().numpy(), g[f"out_{mode}"], 1e-5, 2e-6, f"warp {mode}")
        if mode == "bilinear":
            (out * t(g["grad_out"], cuda)).sum().backward()
            close(x.grad.cpu().numpy(), g["grad_x"], 1e-4, 1e-5, "grad_x")
            close(flow.grad.cpu().numpy(), g["grad_flow"], 1e-4, 1e-5, "grad_flow")
    for scale in (False, True):
        mg = imgflowarp.get_spatial_meshgrid(t(g["x"], cuda), scale=scale)
        close(mg.cpu().numpy(), g[f"meshgrid_{int(scale)}"], 0, 0, "meshgrid")


def test_occlusion_golden(cuda):
    from handobjectconsist_amd.warping import imgflowarp

    g = np.load(os.path.join(GOLDEN, "warp_occlusion.npz"))
    o1, o2 = imgflowarp.get_occlusion_mask(t(g["mask_flow1"], cuda), t(g["mask_flow2"], cuda),
                                           t(g["flow12"], cuda), t(g["flow21"], cuda))
    close(o1.cpu().numpy(), g["occl1"], 0, 1e-7, "occl1")
    close(o2.cpu().numpy(), g["occl2"], 0, 1e-7, "occl2")
    assert g["occl1"].sum() > 10


@pytest.mark.parametrize("use_backward", [False, True])
@pytest.mark.parametrize("outputs", ["full", "loss"])
def test_pair_consist_golden(cuda, use_backward, outputs):
    from handobjectconsist_amd.optim.pyramidloss import PyramidCriterion
    from handobjectconsist_amd.warping import imgflowarp

    g = np.load(os.path.join(GOLDEN, "warp_pair_consist.npz"))
    tag = f"ub{int(use_backward)}"
    f12 = t(g["flow12"], cuda).requires_grad_(True)
    f21 = t(g["flow21"], cuda).requires_grad_(True)
    loss, masks, warps, diffs = imgflowarp.pair_consist(
        [f12, f21], t(g["image_ref"], cuda), t(g["image"], cuda), t(g["jitter_ref"], cuda), t(g["jitter"], cuda),
        PyramidCriterion("l1"), use_backward=use_backward, outputs=outputs)
    close(loss.detach().cpu().numpy(), g[f"loss_{tag}"], 1e-5, 1e-7, "loss")
    (loss * t(g["grad_loss"], cuda)).sum().backward()
    g12 = f12.grad.cpu().numpy() if f12.grad is not None else np.zeros_like(g["flow12"])
    close(g12, g[f"grad_flow12_{tag}"], 1e-4, 1e-7, "grad_flow12")
    close(f21.grad.cpu().numpy(), g[f"grad_flow21_{tag}"], 1e-4, 1e-7, "grad_flow21")
    if outputs == "loss":
        assert masks is None and warps is None and diffs is None
        return
    for i in (0, 1):
        assert (masks[i]["full_mask"].cpu().numpy() != g[f"full_mask{i + 1}"]).sum() == 0
        assert (masks[i]["warp_mask"].cpu().numpy() != g[f"warp_mask{i + 1}"]).sum() == 0
        assert (masks[i]["flow_mask"].cpu().numpy() != g[f"flow_mask{i + 1}"]).sum() == 0
        close(warps[i].cpu().numpy(), g[f"warp{i + 1}"], 1e-5, 2e-6, "warp")
        close(diffs[i].cpu().numpy(), g[f"diff{i + 1}"], 1e-5, 2e-6, "diff")


def test_pair_consist_generic_criterion_matches_fused(cuda):
    """l2 goes through the composed `warp` path (reference control flow); with l1 both paths agree."""
    from handobjectconsist_amd.optim.pyramidloss import PyramidCriterion
    from handobjectconsist_amd.warping import imgflowarp

    g = np.load(os.path.join(GOLDEN, "warp_pair_consist.npz"))
    args = [t(g[k], cuda) for k in ("image_ref", "image", "jitter_ref", "jitter")]
    flows = [t(g["flow12"], cuda), t(g["flow21"], cuda)]
    fused = imgflowarp.pair_consist(flows, *args, PyramidCriterion("l1"), use_backward=True)

    class L1Composed(PyramidCriterion):  # not recognised by the fused fast path
        level_nb = 1

        def __init__(self):
            self.criterion = lambda a, b: (a - b).abs()

    comp = imgflowarp.pair_consist(flows, *args, L1Composed(), use_backward=True)
    close(comp[0].cpu().numpy(), fused[0].cpu().numpy(), 1e-5, 1e-7, "loss composed vs fused")
    for i in (0, 1):
        assert (comp[1][i]["full_mask"] != fused[1][i]["full_mask"]).sum() == 0
        assert (comp[1][i]["warp_mask"] != fused[1][i]["warp_mask"]).sum() == 0
    l2 = imgflowarp.pair_consist(flows, *args, PyramidCriterion("l2"), use_backward=False)
    assert torch.isfinite(l2[0]).all()


def test_pair_consist_large_matches_oracle(cuda):
    """Full-size (B=8, 256x256) loss + flow gradients vs the numpy oracle; non-multiple-of-256
    pixel count as well (270x480 FPHAB frames, BASELINE config 3)."""
    from handobjectconsist_amd.optim.pyramidloss import PyramidCriterion
    from handobjectconsist_amd.warping import imgflowarp

    for (B, H, Wd, seed) in ((4, 256, 256, 0), (2, 270, 480, 1)):
        rng = np.random.default_rng(seed)
        im_ref, im, jm_ref, jm = synth.random_images(B, H, Wd, seed)
        flows = []
        for _ in range(2):
            f = (rng.standard_normal((B, H, Wd, 2)) * 2).astype(np.float32)
            f[rng.random((B, H, Wd)) < 0.5] = 0
            flows.append(f)
        gl = rng.uniform(0.5, 1.5, (B,)).astype(np.float32)
        ref_loss, ref_masks, _, _, _ = W.pair_consist(flows, im_ref, im, jm_ref, jm, True)
        ref_g = W.pair_consist_grad(flows, im_ref, im, jm_ref, jm, gl, True)
        f12 = t(flows[0], cuda).requires_grad_(True)
        f21 = t(flows[1], cuda).requires_grad_(True)
        loss, masks, _, _ = imgflowarp.pair_consist([f12, f21], t(im_ref, cuda), t(im, cuda), t(jm_ref, cuda),
                                                   t(jm, cuda), PyramidCriterion("l1"), use_backward=True)
        close(loss.detach().cpu().numpy(), ref_loss, 1e-5, 1e-7, "loss")
        for i in (0, 1):
            assert (masks[i]["full_mask"].cpu().numpy() != ref_masks[i]["full_mask"]).sum() == 0
        (loss * t(gl, cuda)).sum().backward()
        close(f12.grad.cpu().numpy(), ref_g[0], 1e-4, 1e-9, "grad_flow12")
        close(f21.grad.cpu().numpy(), ref_g[1], 1e-4, 1e-9, "grad_flow21")


@pytest.mark.parametrize("vertex_color_render,fused_epilogue,fused_vertex_stage",
                         [(True, True, False), (True, True, True), (False, True, False), (True, False, False),
                          (False, False, False)])
def test_opticalflow_chain_matches_oracle(cuda, vertex_color_render, fused_epilogue, fused_vertex_stage, monkeypatch):
    """get_opticalflow (two renders + masks + occlusion + crop) and the pair loss on top of it,
    HIP path vs oracle chain, on the synthetic hand+object scene (non-square crop)."""
    from handobjectconsist_amd.neurender.renderer import Renderer
    from handobjectconsist_amd.optim.pyramidloss import PyramidCriterion
    from handobjectconsist_amd.warping import imgflowarp, opticalflow

    monkeypatch.setattr(opticalflow, "USE_VERTEX_COLOR_RENDER", vertex_color_render)
    monkeypatch.setattr(opticalflow, "USE_FUSED_EPILOGUE", fused_epilogue)
    monkeypatch.setattr(opticalflow, "USE_FUSED_VERTEX_STAGE", fused_vertex_stage)
    calls = []
    real_call = opticalflow._lib.call
    monkeypatch.setattr(opticalflow._lib, "call", lambda name, *a: (calls.append(name), real_call(name, *a))[1])
    B, is_, H, Wd = 2, 128, 96, 128
    s = synth.random_scene(B, seed=21, image_size=is_)
    kw = dict(R=np.eye(3, dtype=np.float32)[None], t=np.zeros((1, 3), np.float32),
              dist_coeffs=np.zeros((1, 5), np.float32), orig_size=is_, image_size=is_, anti_aliasing=False,
              near=0.1, far=100, eps=1e-3)
    ref_flows = W.get_opticalflow(R, [s["verts1"], s["verts2"]], s["faces"], [s["K1"], s["K2"]], kw,
                                  orig_img_size=(Wd, H), ignore_face_idxs=synth.HAND_IGNORE_FACES)
    ren = Renderer(image_size=is_, R=torch.eye(3, device=cuda)[None], t=torch.zeros(1, 3, device=cuda),
                   K=torch.ones(1, 3, 3, device=cuda), orig_size=is_, anti_aliasing=False, fill_back=True, near=0.1,
                   no_light=True, light_intensity_ambient=0.8)  # warpreg.py:40-51
    v1 = t(s["verts1"], cuda).requires_grad_(True)
    flows = opticalflow.get_opticalflow([v1, t(s["verts2"], cuda)], t(s["faces"], cuda),
                                        [t(s["K1"], cuda), t(s["K2"], cuda)], ren, orig_img_size=(Wd, H),
                                        detach_textures=False, detach_renders=True,
                                        ignore_face_idxs=synth.HAND_IGNORE_FACES)
    for i in (0, 1):
        assert flows[i].shape == (B, H, Wd, 2)
        got, ref = flows[i].detach().cpu().numpy(), ref_flows[i]
        assert ((got != 0) != (ref != 0)).sum() == 0, "flow support differs"
        close(got, ref, 1e-4, 1e-5, f"flow{i}")
        assert (ref[..., 0] != 0).sum() > 100
    im_ref, im, jm_ref, jm = synth.random_images(B, H, Wd, 3)
    ref_loss = W.pair_consist(ref_flows, im_ref, im, jm_ref, jm, True)[0]
    loss = imgflowarp.pair_consist(flows, t(im_ref, cuda), t(im, cuda), t(jm_ref, cuda), t(jm, cuda),
                                   PyramidCriterion("l1"), use_backward=True, outputs="loss")[0]
    close(loss.detach().cpu().numpy(), ref_loss, 1e-4, 1e-7, "pair loss on rendered flows")
    loss.sum().backward()
    assert v1.grad is not None and torch.isfinite(v1.grad).all() and v1.grad.abs().sum() > 0
    assert ("mr_flow_vertices_forward" in calls) == fused_vertex_stage
    assert ("mr_flow_vertices_backward" in calls) == fused_vertex_stage
    if fused_vertex_stage:  # both renders of the pair go out as one launch over 2B meshes, forward and backward
        assert calls.count("mr_render_flow_forward") == 1 and calls.count("mr_render_flow_backward") == 1
        assert not {"mr_render_vc_forward", "mr_flow_mask", "mr_flow_finalize_backward", "mr_render_vc_backward"} & set(calls)
        assert calls.count("mr_pair_consist_forward") == 1 and calls.count("mr_pair_consist_backward") == 1
        # detach_textures=True: two separate renders (only the first texture set is detached), same values
        v1d = v1.detach().clone().requires_grad_(True)
        flows_d = opticalflow.get_opticalflow([v1d, t(s["verts2"], cuda)], t(s["faces"], cuda),
                                              [t(s["K1"], cuda), t(s["K2"], cuda)], ren, orig_img_size=(Wd, H),
                                              detach_textures=True, detach_renders=True,
                                              ignore_face_idxs=synth.HAND_IGNORE_FACES)
        for i in (0, 1):
            assert torch.equal(flows_d[i].detach(), flows[i].detach())
        assert not flows_d[0].requires_grad and flows_d[1].requires_grad
        flows_d[1].sum().backward()
        assert torch.isfinite(v1d.grad).all() and v1d.grad.abs().sum() > 0
    # all (render path x epilogue x vertex stage) combinations give the same vertex gradient
    ref_key = "_vgrad_ref"
    if not hasattr(test_opticalflow_chain_matches_oracle, ref_key):
        setattr(test_opticalflow_chain_matches_oracle, ref_key, v1.grad.clone())
    else:
        ref_g = getattr(test_opticalflow_chain_matches_oracle, ref_key)
        close(v1.grad.cpu().numpy(), ref_g.cpu().numpy(), 1e-3, 1e-5 * float(ref_g.abs().max()), "vertex grad across paths")


class _ContainerCopyingWrapper(torch.nn.Module):
    def __init__(self, module):
        super().__init__()
        self.module = module

    @staticmethod
    def _copy(obj):
        if isinstance(obj, dict):
            return {k: _ContainerCopyingWrapper._copy(v) for k, v in obj.items()}
        if isinstance(obj, (list, tuple)):
            return type(obj)(_ContainerCopyingWrapper._copy(v) for v in obj)
        return obj

    def forward(self, *args, **kwargs):
        return self.module(*self._copy(args), **self._copy(kwargs))


@pytest.mark.parametrize("batch_post,batch_encoder", [(True, False), (True, True)])
def test_train_step_batched_frames_match_per_frame(cuda, batch_post, batch_encoder):
    """One optimiser step (data batch + consist pair: encoder, MANO, 2 renders, occlusion, pair loss,
    backward, SGD) with the heads / MANO (and optionally the encoder) run ONCE over the three frames
    == the reference's structure of one pass per frame."""
    from handobjectconsist_amd.models.synthnet import SynthMeshRegNet
    from handobjectconsist_amd.models.warpreg import WarpRegNet
    from handobjectconsist_amd.netscripts import epochpassconsist as E

    res = {}
    for batched in (False, True):
        E.BATCH_POST, E.BATCH_ENCODER = (batch_post, batch_encoder) if batched else (False, False)
        torch.manual_seed(0)
        model = SynthMeshRegNet().to(cuda).eval()
        # like DistributedDataParallel with device_ids, the wrapper hands the module re-built COPIES of the
        # input containers: whatever the prepare pass computes has to come back as a return value
        wrapped = _ContainerCopyingWrapper(model) if batched else model
        pre = WarpRegNet((64, 64), wrapped, lambda_consist=0.5, lambda_data=0.5, criterion="l1", gt_refs=True,
                         use_backward=True, mano_faces=model.mano_layer.th_faces, pair_outputs="loss").to(cuda)
        pre.step_count = 1000
        opt = torch.optim.SGD(model.parameters(), lr=1e-3)
        loader = E.SyntheticConsistLoader(3, 64, seed=0, device=cuda, pool=1)
        try:
            loss, logs = E.train_step(loader.step_batches(0), pre, opt)
        finally:
            E.BATCH_POST, E.BATCH_ENCODER = True, True
        assert all("_features" not in s and "_post" not in s for b in loader.step_batches(0) for s in b["data"])
        assert "warp_consist" in logs and float(logs["warp_consist"].detach()) > 0
        res[batched] = {k: float(v.detach().flatten()[0]) for k, v in logs.items()}
    # the smooth terms agree to rounding (different MIOpen kernels for B and 3B images) ...
    for k in ("recov_joint3d", "recov_objverts3d", "mano_reg_loss"):
        assert abs(res[True][k] - res[False][k]) <= 1e-4 * abs(res[False][k]) + 1e-9, (k, res[True][k], res[False][k])
    # ... the photometric term is piecewise constant in the vertices (pixel coverage, validity masks):
    # last-bit differences in the features may flip single pixels of the 64 x 64 test images
    assert abs(res[True]["warp_consist"] - res[False]["warp_consist"]) <= 0.05 * abs(res[False]["warp_consist"])


def test_encode_frames_matches_per_frame_features(cuda):
    from handobjectconsist_amd.models.synthnet import SynthMeshRegNet

    torch.manual_seed(0)
    model = SynthMeshRegNet().to(cuda).eval()
    samples = [{"image": torch.rand(4, 3, 64, 64, device=cuda) - 0.5} for _ in range(3)]
    with torch.no_grad():
        ref = [model.base_net(s["image"]) for s in samples]
        model.encode_frames(samples)
    for s, r in zip(samples, ref):
        assert s["_features"].shape == r.shape
        assert (s["_features"] - r).abs().max() <= 1e-5 * r.abs().max()
    model.train()
    with pytest.raises(RuntimeError):
        model.encode_frames(samples)


@pytest.mark.parametrize("cam_batched", [False, True])
def test_flow_vertex_stage_matches_torch_ops(cuda, cam_batched):
    """mr_flow_vertices_forward / _backward == batch_proj2d + displacement textures + nr.projection and
    their autograd, with a rotated / translated / distorting camera."""
    from handobjectconsist_amd.neurender import nr_ops
    from handobjectconsist_amd.utils import project
    from handobjectconsist_amd.warping.opticalflow import _FlowVertexStage

    B, is_ = 3, 128
    s = synth.random_scene(B, seed=4, image_size=is_)
    g = torch.Generator().manual_seed(1)
    nb = B if cam_batched else 1
    ang = 0.05 * torch.randn(nb, generator=g)
    Rm = torch.eye(3).repeat(nb, 1, 1)
    Rm[:, 0, 0], Rm[:, 0, 1], Rm[:, 1, 0], Rm[:, 1, 1] = ang.cos(), -ang.sin(), ang.sin(), ang.cos()
    tv = 0.01 * torch.randn(nb, 1, 3, generator=g)
    dist = 0.05 * torch.randn(nb, 5, generator=g)
    Rm, tv, dist = Rm.to(cuda), tv.to(cuda), dist.to(cuda)
    v1 = t(s["verts1"], cuda).requires_grad_(True)
    v2 = t(s["verts2"], cuda).requires_grad_(True)
    K1, K2 = t(s["K1"], cuda), t(s["K2"], cuda)
    ndc, cols = _FlowVertexStage.apply(v1, v2, K1, K2, Rm, tv, dist, is_)  # frame 1 then frame 2, stacked
    assert ndc.shape == cols.shape == (2 * B,) + tuple(v1.shape[1:])
    (ndc1, ndc2), (c12, c21) = (ndc[:B], ndc[B:]), (cols[:B], cols[B:])
    a1, a2 = v1.detach().clone().requires_grad_(True), v2.detach().clone().requires_grad_(True)
    p1, p2 = project.batch_proj2d(a1, K1), project.batch_proj2d(a2, K2)
    r12 = torch.cat([p2 - p1, torch.ones_like(p1[..., :1])], -1)
    r21 = torch.cat([p1 - p2, torch.ones_like(p1[..., :1])], -1)
    rn1 = nr_ops.projection(a1, K1, Rm, tv, dist, is_)
    rn2 = nr_ops.projection(a2, K2, Rm, tv, dist, is_)
    assert not ndc1.requires_grad and not ndc2.requires_grad and c12.requires_grad
    for got, ref, what in ((ndc1, rn1, "ndc1"), (ndc2, rn2, "ndc2"), (c12, r12, "cols12"), (c21, r21, "cols21")):
        close(got.detach().cpu().numpy(), ref.detach().cpu().numpy(), 1e-5, 1e-5, what)
    w12, w21 = torch.randn(c12.shape, generator=g).to(cuda), torch.randn(c21.shape, generator=g).to(cuda)
    ((c12 * w12).sum() + (c21 * w21).sum()).backward()
    ((r12 * w12).sum() + (r21 * w21).sum()).backward()
    for got, ref, what in ((v1.grad, a1.grad, "grad verts1"), (v2.grad, a2.grad, "grad verts2")):
        close(got.cpu().numpy(), ref.cpu().numpy(), 1e-4, 1e-5 * float(ref.abs().max()), what)
    # frame 2 detached (warpbranch first_only): no gradient buffer for it
    v1.grad = None
    _, colsb = _FlowVertexStage.apply(v1, v2.detach(), K1, K2, Rm, tv, dist, is_)
    (colsb[:B] * w12).sum().backward()
    assert v1.grad is not None and torch.isfinite(v1.grad).all()


@pytest.mark.parametrize("B,is_,H,Wd", [(2, 480, 270, 480), (1, 640, 480, 640)])
def test_opticalflow_chain_baseline_config_sizes(cuda, B, is_, H, Wd):
    """BASELINE.json configs 3 (480x270 pairs) and 5 (640x480 renders): the default (fully fused) path --
    vertex stage, two vertex-colour renders, epilogue, occlusion, pair loss and its backward -- against
    the oracle chain at the full raster sizes (non-multiples of the 32 / 64-pixel tiles, non-square crop)."""
    from handobjectconsist_amd.neurender.renderer import Renderer
    from handobjectconsist_amd.optim.pyramidloss import PyramidCriterion
    from handobjectconsist_amd.warping import imgflowarp, opticalflow

    s = synth.random_scene(B, seed=33, image_size=is_)
    kw = dict(R=np.eye(3, dtype=np.float32)[None], t=np.zeros((1, 3), np.float32),
              dist_coeffs=np.zeros((1, 5), np.float32), orig_size=is_, image_size=is_, anti_aliasing=False,
              near=0.1, far=100, eps=1e-3)
    ref_flows = W.get_opticalflow(R, [s["verts1"], s["verts2"]], s["faces"], [s["K1"], s["K2"]], kw,
                                  orig_img_size=(Wd, H), ignore_face_idxs=synth.HAND_IGNORE_FACES)
    ren = Renderer(image_size=is_, R=torch.eye(3, device=cuda)[None], t=torch.zeros(1, 3, device=cuda),
                   K=torch.ones(1, 3, 3, device=cuda), orig_size=is_, anti_aliasing=False, fill_back=True, near=0.1,
                   no_light=True, light_intensity_ambient=0.8)
    v1 = t(s["verts1"], cuda).requires_grad_(True)
    flows = opticalflow.get_opticalflow([v1, t(s["verts2"], cuda)], t(s["faces"], cuda),
                                        [t(s["K1"], cuda), t(s["K2"], cuda)], ren, orig_img_size=(Wd, H),
                                        detach_textures=False, detach_renders=True,
                                        ignore_face_idxs=synth.HAND_IGNORE_FACES)
    for i in (0, 1):
        assert flows[i].shape == (B, H, Wd, 2)
        got, ref = flows[i].detach().cpu().numpy(), ref_flows[i]
        # a vertex projected with a last-bit difference may move one pixel across a face edge at these sizes
        assert ((got != 0) != (ref != 0)).sum() <= 4, "flow support differs"
        same = (got != 0) == (ref != 0)
        close(np.where(same, got, 0), np.where(same, ref, 0), 1e-4, 1e-4, f"flow{i}")
        assert (ref[..., 0] != 0).sum() > 1000
    im_ref, im, jm_ref, jm = synth.random_images(B, H, Wd, 3)
    ref_loss = W.pair_consist(ref_flows, im_ref, im, jm_ref, jm, True)[0]
    loss = imgflowarp.pair_consist(flows, t(im_ref, cuda), t(im, cuda), t(jm_ref, cuda), t(jm, cuda),
                                   PyramidCriterion("l1"), use_backward=True, outputs="loss")[0]
    close(loss.detach().cpu().numpy(), ref_loss, 2e-3, 1e-6, "pair loss on rendered flows")
    loss.sum().backward()
    assert torch.isfinite(v1.grad).all() and v1.grad.abs().sum() > 0


@pytest.mark.parametrize("B,center_idx", [(5, 9), (64, 9), (3, None)])
def test_mano_lbs_hip_matches_torch_layer(cuda, B, center_idx):
    """mr_mano_forward / mr_mano_backward (blend-shape GEMM on the matrix cores) == the PyTorch restatement
    of manopth's ManoLayer.forward and its autograd."""
    from handobjectconsist_amd.models import synthnet

    layer = synthnet.SynthManoLayer(ncomps=15, use_pca=True, center_idx=center_idx).to(cuda)
    g = torch.Generator().manual_seed(B)
    pose = (0.4 * torch.randn(B, 18, generator=g)).to(cuda)
    pose[0, :3] = 0  # a zero axis-angle: the 1e-8 guard of the Rodrigues formula
    beta = torch.randn(B, 10, generator=g).to(cuda)
    p1, b1 = pose.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    p2, b2 = pose.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    v_hip, j_hip = layer(p1, b1)
    v_ref, j_ref = layer.forward_torch(p2, b2)
    assert v_hip.shape == (B, 778, 3) and j_hip.shape == (B, 21, 3)
    scale = float(v_ref.detach().abs().max())
    close(v_hip.detach().cpu().numpy(), v_ref.detach().cpu().numpy(), 1e-5, 1e-5 * scale, "verts")
    close(j_hip.detach().cpu().numpy(), j_ref.detach().cpu().numpy(), 1e-5, 1e-5 * scale, "joints")
    wv, wj = torch.randn(v_ref.shape, generator=g).to(cuda), torch.randn(j_ref.shape, generator=g).to(cuda)
    ((v_hip * wv).sum() + (j_hip * wj).sum()).backward()
    ((v_ref * wv).sum() + (j_ref * wj).sum()).backward()
    for got, ref, what in ((p1.grad, p2.grad, "grad pose"), (b1.grad, b2.grad, "grad betas")):
        close(got.cpu().numpy(), ref.cpu().numpy(), 1e-4, 1e-5 * float(ref.abs().max()), what)
    # only one of the two outputs used downstream
    p3, b3 = pose.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    p4, b4 = pose.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    (layer(p3, b3)[1] * wj).sum().backward()
    (layer.forward_torch(p4, b4)[1] * wj).sum().backward()
    close(p3.grad.cpu().numpy(), p4.grad.cpu().numpy(), 1e-4, 1e-5 * float(p4.grad.abs().max()), "grad pose (joints only)")
    close(b3.grad.cpu().numpy(), b4.grad.cpu().numpy(), 1e-4, 1e-5 * float(b4.grad.abs().max()), "grad betas (joints only)")


@pytest.mark.parametrize("B,center_idx", [(4, 9), (33, 9), (2, None)])
def test_mano_lbs_hip_matches_the_numpy_oracle(cuda, B, center_idx):
    """mr_mano_forward / mr_mano_backward against oracle/mano_ref.py -- manopth's ManoLayer.forward restated joint by
    joint in numpy, OUTSIDE the product (manobranch.py:70-85, 130-136): vertices / joints to 1e-5 of
    their scale against its fp64 evaluation, gradients along random directions to 1e-4 against its central
    differences."""
    from handobjectconsist_amd.models import synthnet
    from oracle import mano_ref as M

    layer = synthnet.SynthManoLayer(ncomps=15, use_pca=True, center_idx=center_idx).to(cuda)
    c = {k: getattr(layer, k).detach().cpu().numpy() for k in
         ("th_v_template", "th_shapedirs", "th_posedirs", "th_J_regressor", "th_weights", "th_comps", "th_hands_mean")}
    g = torch.Generator().manual_seed(100 + B)
    pose = 0.4 * torch.randn(B, 18, generator=g)
    pose[0, :3] = 0  # a zero axis-angle: the 1e-8 guard of the Rodrigues formula
    beta = torch.randn(B, 10, generator=g)
    p, b_ = pose.to(cuda).requires_grad_(True), beta.to(cuda).requires_grad_(True)
    v_hip, j_hip = layer(p, b_)
    v_ref, j_ref = M.mano_forward(c, pose.numpy(), beta.numpy(), center_idx=center_idx)
    scale = float(np.abs(v_ref).max())
    close(v_hip.detach().cpu().numpy(), v_ref, 1e-5, 1e-5 * scale, "verts")
    close(j_hip.detach().cpu().numpy(), j_ref, 1e-5, 1e-5 * scale, "joints")
    wv, wj = torch.randn(v_hip.shape, generator=g), torch.randn(j_hip.shape, generator=g)
    ((v_hip * wv.to(cuda)).sum() + (j_hip * wj.to(cuda)).sum()).backward()
    gp, gb = p.grad.double().cpu(), b_.grad.double().cpu()
    for k in range(6):
        dp, db = torch.randn(pose.shape, generator=g), torch.randn(beta.shape, generator=g)
        if k == 0:
            dp[1:], db[1:] = 0, 0  # sample 0 alone (its root rotation is the guarded zero axis-angle)
        num = M.directional_derivative(c, pose.numpy(), beta.numpy(), wv.numpy(), wj.numpy(), dp.numpy(), db.numpy(),
                                       center_idx=center_idx)
        ana = float((gp * dp.double()).sum() + (gb * db.double()).sum())
        assert abs(ana - num) <= 1e-4 * max(abs(num), 1.0) + 1e-4 * float(gp.abs().max()), (k, ana, num)


def test_mano_layer_has_no_silent_gpu_fallback(cuda):
    """Calls the HIP kernels do not cover raise on a GPU unless the layer was built with torch_variants=True."""
    from handobjectconsist_amd.models import synthnet

    layer = synthnet.SynthManoLayer(ncomps=15, use_pca=True, center_idx=9).to(cuda)
    pose, beta = torch.zeros(2, 18, device=cuda), torch.zeros(2, 10, device=cuda)
    with pytest.raises(RuntimeError, match="no HIP kernel"):
        layer(pose, beta, th_trans=torch.ones(2, 3, device=cuda))
    with pytest.raises(RuntimeError, match="no HIP kernel"):
        layer(pose.double(), beta.double())
    with pytest.raises(RuntimeError, match="no HIP kernel"):
        synthnet.SynthManoLayer(ncomps=15, use_pca=False, center_idx=None).to(cuda)(torch.zeros(2, 48, device=cuda), beta)
    allowed = synthnet.SynthManoLayer(ncomps=15, use_pca=True, center_idx=9, torch_variants=True).to(cuda)
    v, _ = allowed(pose, beta, th_trans=torch.ones(2, 3, device=cuda))
    assert v.shape == (2, 778, 3)


def test_mano_constants_follow_the_buffers(cuda):
    """The HIP path's pre-arranged model constants are rebuilt when the layer's buffers are overwritten
    (load_state_dict) -- no stale blend shapes."""
    from handobjectconsist_amd.models import synthnet

    layer = synthnet.SynthManoLayer(ncomps=15, use_pca=True, center_idx=9).to(cuda)
    pose, beta = 0.3 * torch.randn(4, 18, device=cuda), torch.randn(4, 10, device=cuda)
    v0 = layer(pose, beta)[0].clone()
    state = {k: v.clone() for k, v in layer.state_dict().items()}
    state["th_shapedirs"] = state["th_shapedirs"] * 3
    layer.load_state_dict(state)
    v1 = layer(pose, beta)[0]
    v_ref = layer.forward_torch(pose, beta)[0]
    assert float((v1 - v0).abs().max()) > 1e-3
    close(v1.cpu().numpy(), v_ref.cpu().numpy(), 1e-5, 1e-5 * float(v_ref.abs().max()), "verts after load_state_dict")


@pytest.mark.parametrize("B,Vo", [(3, 1002), (64, 257)])
def test_meshreg_post_hip_matches_torch_ops(cuda, B, Vo, monkeypatch):
    """mr_meshreg_post_forward / _backward == recover_3d_proj x2 + unit conversion + object rotation + both
    projections in PyTorch, values and gradients (any subset of the outputs used)."""
    from handobjectconsist_amd.models import synthnet

    model = synthnet.SynthMeshRegNet().to(cuda).eval()
    g = torch.Generator().manual_seed(B)
    mk = lambda *shape, s=1.0: (s * torch.randn(*shape, generator=g)).to(cuda)
    pose, shape = mk(B, 18, s=0.3), mk(B, 10)
    st, so = mk(B, 3), mk(B, 6)
    so[0, 3:] = 0  # zero rotation: the 1e-8 guard
    K = torch.tensor([[350.0, 0.0, 120.0], [0.0, 350.0, 131.0], [0.0, 0.0, 1.0]]).repeat(B, 1, 1).to(cuda)
    K[:, 0, 0] += mk(B, s=20.0); K[:, 1, 1] = K[:, 0, 0]
    can = mk(B, Vo, 3, s=0.05)
    outs = {}
    for hip in (True, False):
        monkeypatch.setattr(synthnet, "USE_HIP_POST", hip)
        leaves = [t_.clone().requires_grad_(True) for t_ in (pose, shape, st, so)]
        o = model.post_heads(*leaves, K, can, input_res=(256, 240))
        outs[hip] = (o, leaves)
    gg = torch.Generator().manual_seed(1)
    ws = [torch.randn(x.shape, generator=gg).to(cuda) for x in outs[False][0]]
    for a, b_, name in zip(outs[True][0], outs[False][0], ("handverts3d", "joints3d", "joints2d", "objverts3d", "objverts2d")):
        close(a.detach().cpu().numpy(), b_.detach().cpu().numpy(), 1e-5, 1e-5 * float(b_.detach().abs().max()), name)
    for subset in ((0, 1, 2, 3, 4), (0,), (2, 4), (3,)):
        for hip in (True, False):
            o, leaves = outs[hip]
            for l in leaves:
                l.grad = None
            sum((o[k] * ws[k]).sum() for k in subset).backward(retain_graph=True)
        for a, b_, name in zip(outs[True][1], outs[False][1], ("pose", "shape", "scaletrans", "st_obj")):
            ga = a.grad if a.grad is not None else torch.zeros_like(a)
            gb = b_.grad if b_.grad is not None else torch.zeros_like(b_)
            close(ga.cpu().numpy(), gb.cpu().numpy(), 2e-4, 2e-5 * float(gb.abs().max()) + 1e-12, f"grad {name} {subset}")


@pytest.mark.parametrize("B,is_,H,Wd", [(2, 128, 96, 128), (3, 64, 64, 64)])
def test_pair_loss_with_coverage_bytes_equals_dense(cuda, B, is_, H, Wd):
    """The stacked get_opticalflow hands the coverage bytes of its renders to pair_consist, which then does not read
    the flows where nothing was rendered: same losses and flow gradients as the dense call, bit for bit -- also when
    those flow values are poisoned (proof that they are not read)."""
    from handobjectconsist_amd.neurender.renderer import Renderer
    from handobjectconsist_amd.optim.pyramidloss import PyramidCriterion
    from handobjectconsist_amd.warping import imgflowarp, opticalflow

    s = synth.random_scene(B, seed=31, image_size=is_)
    ren = Renderer(image_size=is_, R=torch.eye(3, device=cuda)[None], t=torch.zeros(1, 3, device=cuda),
                   K=torch.ones(1, 3, 3, device=cuda), orig_size=is_, anti_aliasing=False, fill_back=True, near=0.1,
                   no_light=True, light_intensity_ambient=0.8)
    flows = opticalflow.get_opticalflow([t(s["verts1"], cuda), t(s["verts2"], cuda)], t(s["faces"], cuda),
                                        [t(s["K1"], cuda), t(s["K2"], cuda)], ren, orig_img_size=(Wd, H),
                                        detach_textures=False, detach_renders=True,
                                        ignore_face_idxs=synth.HAND_IGNORE_FACES)
    base = flows[0]._base
    coverage, cov_size, noted_version = base._hoc_coverage[:3]
    assert noted_version == base._version
    assert cov_size == is_ and coverage.shape == (2 * B, is_ // 8, is_ // 32, 4)
    im_ref, im, jm_ref, jm = [t(a, cuda) for a in synth.random_images(B, H, Wd, 3)]
    crit = PyramidCriterion(criterion="l1")

    def run(stacked_flows, with_coverage):
        fl = stacked_flows.detach().clone().requires_grad_(True)
        if with_coverage:
            fl._hoc_coverage = (coverage, cov_size, fl._version)
        loss, _, _, _ = imgflowarp.pair_consist([fl[:B], fl[B:]], im_ref, im, jm_ref, jm, crit, use_backward=True,
                                                outputs="loss")
        loss.sum().backward()
        return loss.detach(), fl.grad

    calls = []
    real_call = imgflowarp._lib.call
    imgflowarp._lib.call = lambda name, *a: (calls.append((name, a)), real_call(name, *a))[1]
    try:
        loss_c, grad_c = run(base, True)
    finally:
        imgflowarp._lib.call = real_call
    fwd_args = [a for n, a in calls if n == "mr_pair_consist_forward"][0]
    assert fwd_args[-4].value and fwd_args[-3].value and fwd_args[-2] == is_, "the coverage bytes must reach the C-ABI"
    loss_d, grad_d = run(base, False)
    assert torch.equal(loss_c, loss_d) and torch.equal(grad_c, grad_d)
    assert float(loss_d.abs().sum()) > 0 and float(grad_d.abs().sum()) > 0
    # poison the flow wherever the coverage bytes say "nothing rendered" (image orientation: raster row is_ - 1 - y)
    hit = coverage.view(2 * B, is_ // 8, is_ // 32, 4).cpu().numpy()
    yy, xx = np.mgrid[0:H, 0:Wd]
    ry = is_ - 1 - yy
    covered = hit[:, ry >> 3, xx >> 5, (ry & 7) >> 1] != 0          # [2B, H, Wd]
    poisoned = base.detach().clone()
    poisoned[~torch.from_numpy(covered).to(cuda)] = float("nan")
    assert int((~covered).sum()) > 0
    loss_p, grad_p = run(poisoned, True)
    assert torch.equal(loss_p, loss_d) and torch.equal(grad_p, grad_d)
    # an in-place write into the flows after get_opticalflow invalidates the hand-over (the bytes describe other values)
    assert imgflowarp._coverage_of(base)[0] is coverage
    with torch.no_grad():
        base.mul_(1.0)
    assert imgflowarp._coverage_of(base) == (None, 0)


def test_gradient_bound_travels_from_the_pair_loss_to_the_raster_backward(cuda):
    """mr_pair_consist_backward's grad_max = the per-image maxima of |grad_flow| (bit-exact, both directions);
    mr_render_flow_backward with that bound gives the gradient it finds with its own maximum pass (the bound only sets the
    fixed-point scale); and get_opticalflow -> pair_consist -> backward hands the bound over (spied on the C-ABI call)."""
    from handobjectconsist_amd import _lib
    from handobjectconsist_amd.neurender.renderer import Renderer
    from handobjectconsist_amd.optim.pyramidloss import PyramidCriterion
    from handobjectconsist_amd.warping import imgflowarp, opticalflow

    B, is_ = 3, 96
    s = synth.random_scene(B, seed=8, image_size=is_)
    ren = Renderer(image_size=is_, R=torch.eye(3, device=cuda)[None], t=torch.zeros(1, 3, device=cuda), K=torch.ones(1, 3, 3, device=cuda),
                   orig_size=is_, anti_aliasing=False, fill_back=True, near=0.1, no_light=True)
    im_ref, im, jm_ref, jm = [t(a, cuda) for a in synth.random_images(B, is_, is_, 2)]
    crit = PyramidCriterion("l1")
    res = {}
    for handed in (True, False):
        imgflowarp.PASS_GRADIENT_BOUND = handed
        calls = []
        real = _lib.call
        spy = lambda name, *a: (calls.append((name, a)), real(name, *a))[1]
        imgflowarp._lib.call = opticalflow._lib.call = spy
        try:
            v1 = t(s["verts1"], cuda).requires_grad_(True)
            flows = opticalflow.get_opticalflow([v1, t(s["verts2"], cuda)], t(s["faces"], cuda), [t(s["K1"], cuda), t(s["K2"], cuda)], ren,
                                                orig_img_size=(is_, is_), ignore_face_idxs=synth.HAND_IGNORE_FACES)
            flows[0].retain_grad() if False else None
            loss = imgflowarp.pair_consist(flows, im_ref, im, jm_ref, jm, crit, use_backward=True, outputs="loss")[0]
            loss.sum().backward()
        finally:
            imgflowarp._lib.call = opticalflow._lib.call = real
            imgflowarp.PASS_GRADIENT_BOUND = True
        pair_args = [a for n_, a in calls if n_ == "mr_pair_consist_backward"][0]
        flow_args = [a for n_, a in calls if n_ == "mr_render_flow_backward"][0]
        assert bool(pair_args[-2].value if pair_args[-2] is not None else 0) == handed, "grad_max pointer of the pair backward"
        assert bool(flow_args[-2].value if flow_args[-2] is not None else 0) == handed, "grad_bound pointer of the raster backward"
        res[handed] = v1.grad.clone()
    assert float(res[True].abs().sum()) > 0
    close(res[True].cpu().numpy(), res[False].cpu().numpy(), 1e-5, 1e-6 * float(res[False].abs().max()), "gradient with the handed-over bound")

    # the maxima themselves
    g = torch.Generator().manual_seed(3)
    f12 = ((torch.rand(B, is_, is_, 2, generator=g) - 0.5) * 6).to(cuda) * (torch.rand(B, is_, is_, 1, generator=g) < 0.4).to(cuda)
    f21 = ((torch.rand(B, is_, is_, 2, generator=g) - 0.5) * 6).to(cuda) * (torch.rand(B, is_, is_, 1, generator=g) < 0.4).to(cuda)
    P, st = _lib.ptr, _lib.stream_ptr(cuda)
    f32 = dict(dtype=torch.float32, device=cuda)
    wbytes = int(_lib.load().mr_pair_consist_workspace_bytes(B, is_, is_))
    work, sums = torch.empty((max(wbytes, 16),), dtype=torch.uint8, device=cuda), torch.empty((B, 4), **f32)
    lf, lb = torch.empty((B,), **f32), torch.empty((B,), **f32)
    _lib.call("mr_pair_consist_forward", P(f12), P(f21), P(im_ref), P(im), P(jm_ref), P(jm), 3, P(work), wbytes, P(sums), P(lf), P(lb),
              *([None] * 8), B, is_, is_, 0.99999, None, None, 0, st)
    gl = torch.tensor([1.0, -2.5, 0.3], **f32)
    g12, g21, gmax = torch.empty_like(f12), torch.empty_like(f21), torch.zeros(2 * B, **f32)
    _lib.call("mr_pair_consist_backward", P(f12), P(f21), P(im_ref), P(im), P(jm_ref), P(jm), 3, P(sums), P(gl), P(gl), P(g12), P(g21),
              B, is_, is_, 0.99999, None, None, 0, P(gmax), st)
    want = torch.cat([g12.abs().amax((1, 2, 3)), g21.abs().amax((1, 2, 3))])
    assert float(want.max()) > 0 and torch.equal(gmax, want), (gmax, want)


@pytest.mark.parametrize("B,is_,H,W", [(2, 64, 64, 64), (3, 96, 54, 96), (1, 40, 27, 33)])
def test_occlusion_flow_equals_occlusion_then_finalize(cuda, B, is_, H, W):
    """mr_occlusion_flow == mr_occlusion_mask followed by mr_flow_finalize_forward for both directions (SURVEY Q4
    masks: direction 2 uses a raw, non-binary alpha), bit for bit."""
    from handobjectconsist_amd import _lib

    g = torch.Generator().manual_seed(5)
    P, st = _lib.ptr, _lib.stream_ptr(cuda)
    rgb1 = (torch.randn(B, 3, is_, is_, generator=g) * 2).to(cuda)
    rgb2 = (torch.randn(B, 3, is_, is_, generator=g) * 2).to(cuda)
    m1 = (torch.rand(B, is_, is_, generator=g) < 0.6).float().to(cuda)
    m2 = (torch.rand(B, is_, is_, generator=g) < 0.6).float().to(cuda)
    alpha2 = torch.maximum(m2, (torch.rand(B, is_, is_, generator=g) < 0.2).float().to(cuda) * 0.5)
    new = lambda *s: torch.full(s, float("nan"), dtype=torch.float32, device=cuda)
    o1, o2, o1f, o2f = new(B, is_, is_), new(B, is_, is_), new(B, is_, is_), new(B, is_, is_)
    f12, f21, g12, g21 = new(B, H, W, 2), new(B, H, W, 2), new(B, H, W, 2), new(B, H, W, 2)
    _lib.call("mr_occlusion_mask", P(m1), P(alpha2), P(rgb1), P(rgb2), 3 * is_ * is_, P(m1), P(m2), P(o1), P(o2), B, is_, is_,
              0.03, 0.99999, st)
    _lib.call("mr_flow_finalize_forward", P(rgb1), P(m1), P(m1), P(o1), P(f12), B, is_, H, W, st)
    _lib.call("mr_flow_finalize_forward", P(rgb2), P(m2), P(alpha2), P(o2), P(f21), B, is_, H, W, st)
    _lib.call("mr_occlusion_flow", P(m1), P(alpha2), P(rgb1), P(rgb2), 3 * is_ * is_, P(m1), P(m2), P(o1f), P(o2f), P(g12),
              P(g21), None, None, B, is_, is_, H, W, 0.03, 0.99999, st)
    assert torch.equal(o1, o1f) and torch.equal(o2, o2f)
    assert torch.equal(f12, g12) and torch.equal(f21, g21)
    assert float(f12.abs().sum()) > 0 and float(f21.abs().sum()) > 0


@pytest.mark.parametrize("B,is_,H,Wd,bound", [(3, 256, 256, 256, None), (2, 96, 64, 96, 8), (2, 480, 270, 480, None),
                                              (5, 64, 64, 64, 10 ** 6)])
def test_sparse_warp_half_equals_dense(cuda, monkeypatch, B, is_, H, Wd, bound):
    """The sparse contract of the training path (get_opticalflow(sparse_flows=True) -> pair_consist(outputs="loss")):
    occlusion + epilogue, pair loss and pair-loss backward run over the render's tile list and WRITE under the covered
    tiles only.  On NaN-poisoned buffers: the flows equal the dense path's bit for bit under every tile with a non-zero
    coverage word and are untouched elsewhere; losses agree to fp32 rounding (the per-tile partial sums are added in
    another fixed order); the gradient that reaches the vertices of both frames agrees -- i.e. the raster backward read
    the flow gradient and the occlusion mask nowhere outside the coverage.  `bound`: the tile-list guess (8 = nearly the
    whole list goes through the kernels' grid-stride rounds; None: 'auto' first, then the previous call's length)."""
    from handobjectconsist_amd.neurender.renderer import Renderer
    from handobjectconsist_amd.optim.pyramidloss import PyramidCriterion
    from handobjectconsist_amd.warping import imgflowarp, opticalflow

    s = synth.random_scene(B, seed=7, image_size=is_)
    ren = Renderer(image_size=is_, R=torch.eye(3, device=cuda)[None], t=torch.zeros(1, 3, device=cuda),
                   K=torch.ones(1, 3, 3, device=cuda), orig_size=is_, anti_aliasing=False, fill_back=True, near=0.1,
                   no_light=True, light_intensity_ambient=0.8)
    im_ref, im, jm_ref, jm = [t(a, cuda) for a in synth.random_images(B, H, Wd, 5)]
    crit = PyramidCriterion(criterion="l1")
    weights = torch.linspace(0.5, 1.5, B, device=cuda)
    monkeypatch.setattr(opticalflow, "DEBUG_POISON_RENDER_OUTPUTS", True)
    monkeypatch.setattr(imgflowarp, "DEBUG_POISON_SPARSE_GRADS", True)
    if bound is not None:
        real_bound = opticalflow._tile_bound
        monkeypatch.setattr(opticalflow, "_tile_bound", lambda dev, B2, size: (bound, real_bound(dev, B2, size)[1]))
    calls = []
    real_call = imgflowarp._lib.call
    monkeypatch.setattr(imgflowarp._lib, "call", lambda name, *a: (calls.append(name), real_call(name, *a))[1])

    def run(sparse):
        v1, v2 = t(s["verts1"], cuda).requires_grad_(True), t(s["verts2"], cuda).requires_grad_(True)
        flows = opticalflow.get_opticalflow([v1, v2], t(s["faces"], cuda), [t(s["K1"], cuda), t(s["K2"], cuda)], ren,
                                            orig_img_size=(Wd, H), detach_textures=False, detach_renders=True,
                                            ignore_face_idxs=synth.HAND_IGNORE_FACES, sparse_flows=sparse)
        base = flows[0]._base
        loss, masks, _, _ = imgflowarp.pair_consist(flows, im_ref, im, jm_ref, jm, crit, use_backward=True, outputs="loss")
        assert masks is None
        (loss * weights).sum().backward()
        torch.cuda.synchronize()
        return base.detach().clone(), base._hoc_coverage, loss.detach().clone(), v1.grad.clone(), v2.grad.clone()

    flow_d, note_d, loss_d, g1_d, g2_d = run(False)
    assert note_d[3] is None and "mr_occlusion_flow_tiles" not in calls
    assert float(loss_d.abs().sum()) > 0 and float(g1_d.abs().sum()) > 0 and torch.isfinite(flow_d).all()
    for _ in range(2):  # (the second call takes the first one's list length as its guess)
        del calls[:]
        flow_s, note_s, loss_s, g1_s, g2_s = run(True)
        for name in ("mr_occlusion_flow_tiles", "mr_pair_consist_forward_tiles", "mr_pair_consist_backward_tiles"):
            assert name in calls, f"{name} was not launched: {calls}"
        assert not any(n in calls for n in ("mr_occlusion_flow", "mr_pair_consist_forward", "mr_pair_consist_backward"))
        assert note_s[3] is not None and torch.equal(note_s[0], note_d[0])
        # per pixel of the crop: is the coverage WORD of its tile non-zero?  (image orientation: raster row is_ - 1 - y)
        words = note_s[0].contiguous().view(torch.int32).view(2 * B, (is_ + 7) // 8, (is_ + 31) // 32).cpu().numpy() != 0
        yy, xx = np.mgrid[0:H, 0:Wd]
        defined = torch.from_numpy(words[:, (is_ - 1 - yy) >> 3, xx >> 5]).to(cuda)  # [2B, H, Wd]
        assert 0 < int(defined.sum()) < defined.numel()
        assert torch.equal(flow_s[defined], flow_d[defined]), "flows under the covered tiles"
        assert torch.isnan(flow_s[~defined]).all(), "something was written outside the covered tiles"
        close(loss_s.cpu().numpy(), loss_d.cpu().numpy(), 2e-6, 1e-9, "pair loss")
        for a, b_, what in ((g1_s, g1_d, "d/d vertices of frame 1"), (g2_s, g2_d, "d/d vertices of frame 2")):
            assert torch.isfinite(a).all(), what
            close(a.cpu().numpy(), b_.cpu().numpy(), 1e-5, 1e-6 * float(b_.abs().max()), what)
    # flows that are defined under the coverage only must not reach the per-pixel ("full") outputs
    flows = opticalflow.get_opticalflow([t(s["verts1"], cuda), t(s["verts2"], cuda)], t(s["faces"], cuda),
                                        [t(s["K1"], cuda), t(s["K2"], cuda)], ren, orig_img_size=(Wd, H), sparse_flows=True)
    with pytest.raises(ValueError, match="sparse_flows"):
        imgflowarp.pair_consist(flows, im_ref, im, jm_ref, jm, crit, use_backward=True, outputs="full")


@pytest.mark.parametrize("unit", [True, False])
@pytest.mark.parametrize("B,is_,H,Wd,bound", [(3, 256, 256, 256, None), (2, 96, 64, 96, 8), (2, 480, 270, 480, None),
                                              (5, 64, 64, 64, 10 ** 6)])
def test_fused_pair_node_equals_the_composed_path(cuda, monkeypatch, B, is_, H, Wd, bound, unit):
    """opticalflow.flow_pair_loss -- render, ONE pass for occlusion + epilogue + pair loss (mr_flow_pair_forward_tiles), ONE
    backward launch for the pair loss's backward + the epilogue's adjoint + the scatter to the vertices
    (mr_flow_pair_backward_tiles) -- against get_opticalflow(sparse_flows=True) -> pair_consist(outputs="loss"), the same
    arithmetic in separate launches: losses and flows (under the covered tiles) bit for bit, vertex gradients of both frames
    to fp32 rounding of the per-workgroup sums; on NaN-poisoned buffers, with per-sample loss weights on both terms.
    ``unit``: the forward launch leaves the pair loss's gradient for a unit coefficient (mr_flow_pair_forward_grad_tiles) and
    the backward launch is the scatter alone (mr_flow_pair_backward_unit_tiles) -- what training runs."""
    from handobjectconsist_amd.neurender.renderer import Renderer
    from handobjectconsist_amd.optim.pyramidloss import PyramidCriterion
    from handobjectconsist_amd.warping import imgflowarp, opticalflow

    s = synth.random_scene(B, seed=9, image_size=is_)
    ren = Renderer(image_size=is_, R=torch.eye(3, device=cuda)[None], t=torch.zeros(1, 3, device=cuda),
                   K=torch.ones(1, 3, 3, device=cuda), orig_size=is_, anti_aliasing=False, fill_back=True, near=0.1,
                   no_light=True, light_intensity_ambient=0.8)
    im_ref, im, jm_ref, jm = [t(a, cuda) for a in synth.random_images(B, H, Wd, 6)]
    crit = PyramidCriterion(criterion="l1")
    wf, wb = torch.linspace(0.5, 1.5, B, device=cuda), torch.linspace(2.0, 0.25, B, device=cuda)
    monkeypatch.setattr(opticalflow, "DEBUG_POISON_RENDER_OUTPUTS", True)
    monkeypatch.setattr(imgflowarp, "DEBUG_POISON_SPARSE_GRADS", True)
    monkeypatch.setattr(opticalflow, "USE_UNIT_GRADIENT", unit)
    if bound is not None:
        real_bound = opticalflow._tile_bound
        monkeypatch.setattr(opticalflow, "_tile_bound", lambda dev, B2, size: (bound, real_bound(dev, B2, size)[1]))
    calls = []
    real_call = imgflowarp._lib.call
    monkeypatch.setattr(imgflowarp._lib, "call", lambda name, *a: (calls.append(name), real_call(name, *a))[1])
    args = lambda v1, v2: ([v1, v2], t(s["faces"], cuda), [t(s["K1"], cuda), t(s["K2"], cuda)], ren)  # noqa: E731

    def composed(only_fwd):
        v1, v2 = t(s["verts1"], cuda).requires_grad_(True), t(s["verts2"], cuda).requires_grad_(True)
        flows = opticalflow.get_opticalflow(*args(v1, v2), orig_img_size=(Wd, H), detach_textures=False, detach_renders=True,
                                            ignore_face_idxs=synth.HAND_IGNORE_FACES, sparse_flows=True)
        res = imgflowarp._PairConsistFunction.apply(flows[0]._base, None, im_ref, im, jm_ref, jm, 0.99999, False,
                                                    *imgflowarp._coverage_of(flows[0]._base), imgflowarp._tiles_of(flows[0]._base))
        lf, lb = res[0], res[1]
        ((lf * wf).sum() if only_fwd else (lf * wf).sum() + (lb * wb).sum()).backward()
        return lf.detach(), lb.detach(), flows[0]._base.detach().clone(), flows[0]._base._hoc_coverage[0], v1.grad, v2.grad

    def fused(only_fwd):
        v1, v2 = t(s["verts1"], cuda).requires_grad_(True), t(s["verts2"], cuda).requires_grad_(True)
        res = opticalflow.flow_pair_loss(*args(v1, v2), (Wd, H), im_ref, im, jm_ref, jm, ignore_face_idxs=synth.HAND_IGNORE_FACES)
        assert res is not None, "the fused node must apply to this configuration"
        lf, lb, flows = res
        ((lf * wf).sum() if only_fwd else (lf * wf).sum() + (lb * wb).sum()).backward()
        return lf.detach(), lb.detach(), flows[0]._base.detach().clone(), flows[0]._base._hoc_coverage[0], v1.grad, v2.grad

    for only_fwd in (False, True):
        ref = composed(only_fwd)
        del calls[:]
        got = fused(only_fwd)
        if unit:
            assert "mr_flow_pair_forward_grad_tiles" in calls and "mr_flow_pair_backward_unit_tiles" in calls
            assert "mr_flow_pair_forward_tiles" not in calls and "mr_flow_pair_backward_tiles" not in calls
        else:
            assert "mr_flow_pair_forward_tiles" in calls and "mr_flow_pair_backward_tiles" in calls
        assert not any(n in calls for n in ("mr_pair_consist_backward_tiles", "mr_render_flow_backward", "mr_occlusion_flow_tiles"))
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), "losses"
        assert float(ref[0].abs().sum()) > 0 and float(ref[1].abs().sum()) > 0
        assert torch.equal(got[3], ref[3]), "coverage bytes"
        words = ref[3].contiguous().view(torch.int32).view(2 * B, (is_ + 7) // 8, (is_ + 31) // 32).cpu().numpy() != 0
        yy, xx = np.mgrid[0:H, 0:Wd]
        defined = torch.from_numpy(words[:, (is_ - 1 - yy) >> 3, xx >> 5]).to(cuda)
        assert torch.equal(got[2][defined], ref[2][defined]) and torch.isnan(got[2][~defined]).all(), "flows"
        for a, b_, what in ((got[4], ref[4], "d/d vertices of frame 1"), (got[5], ref[5], "d/d vertices of frame 2")):
            assert torch.isfinite(a).all() and float(b_.abs().sum()) > 0, what
            close(a.cpu().numpy(), b_.cpu().numpy(), 1e-5, 1e-6 * float(b_.abs().max()), what)


@pytest.mark.parametrize("B,is_,shape", [(4, 256, "unbalanced"), (1, 64, "unbalanced"), (6, 256, "plain"), (3, 480, "unbalanced"),
                                         (2, 640, "plain")])
def test_scatter_work_lists_equal_the_listing_form(cuda, monkeypatch, B, is_, shape):
    """mr_flow_pair_backward_unit_tiles over the covered-tile lists of the forward's finalize launch (ABI 7: workgroups handed out
    in proportion to the images' covered tiles) against its listing form (a fixed number of workgroups per image, each
    compacting the coverage words itself): the same vertex gradients up to the order of the final fp32 atomics.  "unbalanced":
    one pair zoomed until the meshes fill the screen (hundreds of covered tiles, several rounds per wave), one pushed off
    screen (no covered tile at all, in either frame).  The list buffer comes back full of 0xff bytes: nothing of it may be
    assumed initialised."""
    from handobjectconsist_amd.neurender.renderer import Renderer
    from handobjectconsist_amd.warping import opticalflow

    s = synth.random_scene(B, seed=77, image_size=is_)
    if shape == "unbalanced":
        for k in ("K1", "K2"):
            s[k][0, 0, 0] *= 5.0
            s[k][0, 1, 1] *= 5.0
        if B > 1:
            for k in ("verts1", "verts2"):
                s[k][B - 1, :, 0] += 50.0
    ren = Renderer(image_size=is_, R=torch.eye(3, device=cuda)[None], t=torch.zeros(1, 3, device=cuda),
                   K=torch.ones(1, 3, 3, device=cuda), orig_size=is_, anti_aliasing=False, fill_back=True, near=0.1,
                   no_light=True, light_intensity_ambient=0.8)
    im_ref, im, jm_ref, jm = [t(a, cuda) for a in synth.random_images(B, is_, is_, 4)]
    wf, wb = torch.linspace(0.5, 1.5, B, device=cuda), torch.linspace(2.0, 0.25, B, device=cuda)
    real_empty = torch.empty

    def dirty_empty(*a, **k):  # uint8 scratch buffers come back full of 0xff
        out = real_empty(*a, **k)
        if out.dtype == torch.uint8 and out.numel() > 64:
            out.fill_(255)
        return out

    monkeypatch.setattr(torch, "empty", dirty_empty)
    calls = []
    real_call = opticalflow._lib.call
    monkeypatch.setattr(opticalflow._lib, "call", lambda name, *a: (calls.append((name, a)), real_call(name, *a))[1])

    def run(work):
        monkeypatch.setattr(opticalflow, "USE_SCATTER_WORK", work)
        v1, v2 = t(s["verts1"], cuda).requires_grad_(True), t(s["verts2"], cuda).requires_grad_(True)
        res = opticalflow.flow_pair_loss([v1, v2], t(s["faces"], cuda), [t(s["K1"], cuda), t(s["K2"], cuda)], ren, (is_, is_),
                                         im_ref, im, jm_ref, jm, ignore_face_idxs=synth.HAND_IGNORE_FACES)
        assert res is not None
        lf, lb, _ = res
        del calls[:]
        ((lf * wf).sum() + (lb * wb).sum()).backward()
        bwd = [a for n_, a in calls if n_ == "mr_flow_pair_backward_unit_tiles"]
        assert len(bwd) == 1 and (bwd[0][-2] is not None) == work, "the scatter_work argument of the backward call"
        return lf.detach(), lb.detach(), v1.grad, v2.grad

    ref = run(False)
    got = run(True)
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    for a, b_ in ((got[2], ref[2]), (got[3], ref[3])):
        assert torch.isfinite(a).all() and float(b_.abs().sum()) > 0
        close(a.cpu().numpy(), b_.cpu().numpy(), 1e-5, 2e-6 * float(b_.abs().max()), "vertex gradient")
        if shape == "unbalanced" and B > 1:
            assert float(a[B - 1].abs().sum()) == 0.0, "an off-screen mesh has no gradient"


@pytest.mark.parametrize("B,is_,H,Wd,Cj,batched_hand,per_sample_cam", [
    (3, 256, 256, 256, 3, False, False), (2, 96, 64, 80, 1, True, False), (1, 480, 270, 480, 3, False, False),
    (5, 128, 128, 128, 1, False, False), (3, 128, 128, 96, 3, False, True)])
def test_pair_step_equals_the_node_pair(cuda, monkeypatch, B, is_, H, Wd, Cj, batched_hand, per_sample_cam):
    """ABI 8: flow_pair_loss on (hand, object) parts through mr_pair_step_forward / _backward (ONE struct call each way, one
    autograd node, scratch kept by the plan: warping/pairstep.py) against the node pair it replaces on the host
    (_FlowVertexStageParts + _FlowPairLossFunction, five calls): the same launches, so losses, flows under the covered tiles
    and coverage bytes bit for bit; the vertex gradients of both parts to the order of the backward's fp32 atomics; the batch
    mean against torch.mean of the per-sample losses.  NaN / 0xff-poisoned buffers; two passes through the SAME plan (the scratch
    of the first pass is reused by the second); gradients through loss_fwd, loss_bwd, their sum and the mean at once; a second
    backward through one forward call (retain_graph: the gradient buffer is no longer the one the forward cleared)."""
    from handobjectconsist_amd.neurender.renderer import Renderer
    from handobjectconsist_amd.warping import opticalflow, pairstep

    ren = Renderer(image_size=is_, R=torch.eye(3, device=cuda)[None], t=torch.zeros(1, 3, device=cuda),
                   K=torch.ones(1, 3, 3, device=cuda), orig_size=is_, anti_aliasing=False, fill_back=True, near=0.1,
                   no_light=True, light_intensity_ambient=0.8)
    if per_sample_cam:  # (R / t / dist_coeffs with a batch dimension of B: small rotations about z, shifts, a little distortion)
        ang = torch.linspace(-0.05, 0.05, B, device=cuda)
        R = torch.eye(3, device=cuda).repeat(B, 1, 1)
        R[:, 0, 0], R[:, 0, 1], R[:, 1, 0], R[:, 1, 1] = torch.cos(ang), -torch.sin(ang), torch.sin(ang), torch.cos(ang)
        ren.R, ren.t = R, torch.linspace(-0.01, 0.01, B, device=cuda)[:, None].repeat(1, 3).contiguous()
        ren.dist_coeffs = torch.linspace(-0.02, 0.02, B, device=cuda)[:, None] * torch.tensor([1.0, 0.5, 0.1, -0.1, 0.2], device=cuda)
    monkeypatch.setattr(opticalflow, "DEBUG_POISON_RENDER_OUTPUTS", True)
    wf, wb = torch.linspace(0.5, 1.5, B, device=cuda), torch.linspace(2.0, 0.25, B, device=cuda)
    ws = torch.linspace(-0.5, 0.75, B, device=cuda)
    steps = []
    real_step = pairstep.pair_step
    monkeypatch.setattr(pairstep, "pair_step", lambda *a, **k: (steps.append(1), real_step(*a, **k))[1])

    def run(seed, step, mean_kind="sum", twice=False):
        monkeypatch.setattr(opticalflow, "USE_PAIR_STEP", step)
        s = synth.random_scene(B, seed=seed, image_size=is_)
        im_ref, im, jm_ref, jm = [t(a, cuda) for a in synth.random_images(B, H, Wd, seed + 1)]
        jm_ref, jm = jm_ref[:, :Cj].contiguous(), jm[:, :Cj].contiguous()
        hand_faces = t(s["hand_faces"].astype(np.int64), cuda)
        if batched_hand:
            hand_faces = hand_faces[None].repeat(B, 1, 1)
        obj_faces = t(s["obj_faces"].astype(np.int64)[None].repeat(B, 0), cuda)
        leaves = [t(s[k], cuda).requires_grad_(True) for k in ("hand_verts1", "obj_verts1", "hand_verts2", "obj_verts2")]
        res = opticalflow.flow_pair_loss([(leaves[0], leaves[1]), (leaves[2], leaves[3])], (hand_faces, obj_faces),
                                         [t(s["K1"], cuda), t(s["K2"], cuda)], ren, (Wd, H), im_ref, im, jm_ref, jm,
                                         ignore_face_idxs=synth.HAND_IGNORE_FACES, with_sum=True, with_mean=mean_kind)
        assert res is not None
        lf, lb, flows, lsum, mean = res
        total = (lf * wf).sum() + (lb * wb).sum() + (lsum * ws).sum() + 3.0 * mean
        grads = torch.autograd.grad(total, leaves, retain_graph=twice)
        if twice:
            again = torch.autograd.grad(total, leaves)
            for g, g2 in zip(grads, again):
                close(g2.cpu().numpy(), g.cpu().numpy(), 1e-5, 1e-6 * float(g.abs().max()) + 1e-30, "second backward through one forward")
        base = flows[0]._base
        hit = base._hoc_coverage[0]
        words = hit.contiguous().view(torch.int32).view(2 * B, (is_ + 7) // 8, (is_ + 31) // 32).cpu().numpy() != 0
        yy, xx = np.mgrid[0:H, 0:Wd]
        defined = torch.from_numpy(words[:, (is_ - 1 - yy) >> 3, xx >> 5]).to(cuda)
        assert torch.isnan(base.detach()[~defined]).all(), "nothing is written under uncovered tiles"
        return lf.detach(), lb.detach(), lsum.detach(), base.detach()[defined], hit.clone(), mean.detach(), grads

    for seed, mean_kind, twice in ((31, "sum", False), (32, "fwd", True)):
        del steps[:]
        got = run(seed, True, mean_kind, twice)
        assert steps, "the struct path must have been taken"
        del steps[:]
        ref = run(seed, False, mean_kind)
        assert not steps
        for x, y, what in zip(got[:5], ref[:5], ("loss_fwd", "loss_bwd", "loss_bwd + loss_fwd", "flows", "coverage bytes")):
            assert torch.equal(x, y), (what, seed)
        assert float(ref[0].abs().sum()) > 0
        assert abs(float(got[5]) - float(ref[5])) <= 2e-6 * max(abs(float(ref[5])), 1e-12), "batch mean"
        for x, y, what in zip(got[6], ref[6], ("hand 1", "object 1", "hand 2", "object 2")):
            assert torch.isfinite(x).all() and float(y.abs().sum()) > 0, what
            close(x.cpu().numpy(), y.cpu().numpy(), 1e-5, 1e-6 * float(y.abs().max()), "d/d vertices of " + what)


@pytest.mark.parametrize("B,is_", [(2, 96), (3, 256), (1, 480)])
def test_per_face_pass_inside_the_binning_pass(cuda, monkeypatch, B, is_):
    """flow_pair_loss on (hand, object) parts: the pair prologue clears the header of the render's tile list and the render
    (MR_FLAG_TILE_LIST_CLEARED) computes the face boxes inside its binning kernel instead of a launch of their own -- same
    losses, flows and coverage bytes bit for bit (vertex gradients to the order of the backward's fp32 atomics) as with the
    separate per-face pass; the workspace is
    filled with 0xff bytes first (the list counters must come from the prologue, not from whatever the allocation held)."""
    from handobjectconsist_amd import _lib
    from handobjectconsist_amd.neurender.renderer import Renderer
    from handobjectconsist_amd.warping import opticalflow

    s = synth.random_scene(B, seed=31, image_size=is_)
    ren = Renderer(image_size=is_, R=torch.eye(3, device=cuda)[None], t=torch.zeros(1, 3, device=cuda),
                   K=torch.ones(1, 3, 3, device=cuda), orig_size=is_, anti_aliasing=False, fill_back=True, near=0.1,
                   no_light=True, light_intensity_ambient=0.8)
    im_ref, im, jm_ref, jm = [t(a, cuda) for a in synth.random_images(B, is_, is_, 9)]
    Ks = [t(s["K1"], cuda), t(s["K2"], cuda)]
    hand_faces = t(s["hand_faces"].astype(np.int64), cuda)
    obj_faces = t(s["obj_faces"].astype(np.int64)[None].repeat(B, 0), cuda)
    real_empty = torch.empty

    def dirty_empty(*a, **k):  # uint8 workspaces come back full of 0xff
        out = real_empty(*a, **k)
        if out.dtype == torch.uint8 and out.numel() > 4096:
            out.fill_(255)
        return out

    flags_seen = []
    real_call = _lib.call

    def spy(name, *a):
        if name == "mr_render_flow_forward":  # (`flags` follows the last float argument, eps)
            flags_seen.append(a[max(i for i, x in enumerate(a) if isinstance(x, float)) + 1])
        return real_call(name, *a)

    monkeypatch.setattr(_lib, "call", spy)

    monkeypatch.setattr(opticalflow, "USE_PAIR_STEP", False)  # (the node pair's five calls: the struct path has its own test below)

    def run(fused_records):
        monkeypatch.setattr(opticalflow, "USE_FUSED_RECORDS", fused_records)
        monkeypatch.setattr(torch, "empty", dirty_empty)
        try:
            h1, o1 = t(s["hand_verts1"], cuda).requires_grad_(True), t(s["obj_verts1"], cuda).requires_grad_(True)
            res = opticalflow.flow_pair_loss([(h1, o1), (t(s["hand_verts2"], cuda), t(s["obj_verts2"], cuda))], (hand_faces, obj_faces),
                                             Ks, ren, (is_, is_), im_ref, im, jm_ref, jm, ignore_face_idxs=synth.HAND_IGNORE_FACES)
            assert res is not None
            (res[0] * 1.5 + res[1]).sum().backward()
        finally:
            monkeypatch.setattr(torch, "empty", real_empty)
        base = res[2][0]._base
        hit = base._hoc_coverage[0]
        words = hit.contiguous().view(torch.int32).view(2 * B, (is_ + 7) // 8, (is_ + 31) // 32).cpu().numpy() != 0
        yy, xx = np.mgrid[0:is_, 0:is_]
        defined = torch.from_numpy(words[:, (is_ - 1 - yy) >> 3, xx >> 5]).to(cuda)
        return res[0].detach(), res[1].detach(), base.detach()[defined], hit.clone(), h1.grad, o1.grad

    a, b_ = run(True), run(False)
    assert len(flags_seen) == 2 and flags_seen[0] & _lib.FLAG_TILE_LIST_CLEARED and not flags_seen[1] & _lib.FLAG_TILE_LIST_CLEARED
    for x, y, what in zip(a[:4], b_[:4], ("loss_fwd", "loss_bwd", "flows", "coverage bytes")):
        assert torch.equal(x, y), what
    for x, y, what in zip(a[4:], b_[4:], ("d/d hand vertices", "d/d object vertices")):
        # (the eight workgroups of an image add their table sums to the vertex rows with fp32 atomics: order-dependent last bits)
        close(x.cpu().numpy(), y.cpu().numpy(), 1e-5, 1e-6 * float(y.abs().max()), what)
    assert float(a[0].abs().sum()) > 0 and float(a[4].abs().sum()) > 0


@pytest.mark.parametrize("B,batched_hand", [(1, False), (3, False), (2, True)])
def test_pair_prologue_equals_its_two_launches(cuda, B, batched_hand):
    """mr_flow_pair_prologue_parts (vertex stage of the (hand, object) parts + the stacked int32 faces in ONE launch, what
    flow_pair_loss issues) against mr_flow_vertices_parts_forward and mr_stack_pair_faces: bit for bit."""
    from handobjectconsist_amd.warping import opticalflow

    s = synth.random_scene(B, seed=23, image_size=96)
    Ks = [t(s["K1"], cuda), t(s["K2"], cuda)]
    cam = (Ks[0], Ks[1], torch.eye(3, device=cuda)[None], torch.zeros(1, 3, device=cuda), torch.zeros(1, 5, device=cuda), 96)
    parts = [t(s[k], cuda) for k in ("hand_verts1", "obj_verts1", "hand_verts2", "obj_verts2")]
    hand_faces = t(s["hand_faces"].astype(np.int64), cuda)
    if batched_hand:
        hand_faces = hand_faces[None].repeat(B, 1, 1).contiguous()
    obj_faces = t(s["obj_faces"].astype(np.int64)[None].repeat(B, 0), cuda)
    ndc0, cols0 = opticalflow._FlowVertexStageParts.apply(*parts, *cam)
    faces0 = opticalflow._stack_pair_faces(hand_faces, obj_faces, parts[0].shape[1])
    ndc1, cols1, faces1 = opticalflow._FlowVertexStageParts.apply(*parts, *cam, hand_faces, obj_faces)
    assert torch.equal(ndc0, ndc1) and torch.equal(cols0, cols1) and torch.equal(faces0, faces1)
    want = np.concatenate([np.broadcast_to(s["hand_faces"], (B,) + s["hand_faces"].shape),
                           np.broadcast_to(s["obj_faces"] + parts[0].shape[1], (B,) + s["obj_faces"].shape)], 1)
    assert faces1.dtype == torch.int32 and np.array_equal(faces1.cpu().numpy(), np.concatenate([want, want], 0))


@pytest.mark.parametrize("B,is_,H,Wd,Cj", [(1, 72, 72, 72, 1), (3, 104, 56, 100, 1), (2, 136, 136, 136, 3)])
def test_fused_pair_node_on_ragged_rasters_and_as_parts(cuda, B, is_, H, Wd, Cj):
    """Rasters that are no multiple of the 32 x 8 tile (border tiles hang over the raster), a single pair, crops narrower
    than the raster, one-channel jitter masks -- and the mesh handed over as (hand, object) vertex tensors + (hand, object)
    faces: concatenation and index offset inside the kernels (mr_flow_vertices_parts_*, mr_stack_pair_faces) must give what
    torch.cat of the parts gives, bit for bit (losses, flows) / to fp32 rounding (gradients of BOTH parts of frame 1).
    The composed dense path is the reference throughout."""
    from handobjectconsist_amd.neurender.renderer import Renderer
    from handobjectconsist_amd.optim.pyramidloss import PyramidCriterion
    from handobjectconsist_amd.warping import imgflowarp, opticalflow

    s = synth.random_scene(B, seed=17, image_size=is_)
    ren = Renderer(image_size=is_, R=torch.eye(3, device=cuda)[None], t=torch.zeros(1, 3, device=cuda),
                   K=torch.ones(1, 3, 3, device=cuda), orig_size=is_, anti_aliasing=False, fill_back=True, near=0.1,
                   no_light=True, light_intensity_ambient=0.8)
    im_ref, im, jm_ref, jm = [t(a, cuda) for a in synth.random_images(B, H, Wd, 8)]
    jm_ref, jm = jm_ref[:, :Cj].contiguous(), jm[:, :Cj].contiguous()
    Ks = [t(s["K1"], cuda), t(s["K2"], cuda)]
    Vh = s["hand_verts1"].shape[1]
    hand_faces = t(s["hand_faces"].astype(np.int64), cuda)                              # [Fh,3], shared by the batch
    obj_faces = t(s["obj_faces"].astype(np.int64)[None].repeat(B, 0), cuda)             # [B,Fo,3] WITHOUT the hand offset
    assert int(hand_faces.max()) < Vh and int(obj_faces.min()) >= 0

    def dense():
        v1 = t(s["verts1"], cuda).requires_grad_(True)
        flows = opticalflow.get_opticalflow([v1, t(s["verts2"], cuda)], t(s["faces"], cuda), Ks, ren, orig_img_size=(Wd, H),
                                            ignore_face_idxs=synth.HAND_IGNORE_FACES)
        loss = imgflowarp.pair_consist(flows, im_ref, im, jm_ref, jm, PyramidCriterion("l1"), use_backward=True, outputs="loss")[0]
        loss.sum().backward()
        return loss.detach(), flows[0]._base.detach().clone(), v1.grad

    def fused(parts):
        if parts:
            h1, o1 = t(s["hand_verts1"], cuda).requires_grad_(True), t(s["obj_verts1"], cuda).requires_grad_(True)
            verts = [(h1, o1), (t(s["hand_verts2"], cuda), t(s["obj_verts2"], cuda))]
            faces = (hand_faces, obj_faces)
        else:
            v1 = t(s["verts1"], cuda).requires_grad_(True)
            verts, faces = [v1, t(s["verts2"], cuda)], t(s["faces"], cuda)
        res = opticalflow.flow_pair_loss(verts, faces, Ks, ren, (Wd, H), im_ref, im, jm_ref, jm, ignore_face_idxs=synth.HAND_IGNORE_FACES)
        assert res is not None
        (res[0] + res[1]).sum().backward()
        grad = torch.cat([h1.grad, o1.grad], 1) if parts else v1.grad
        return (res[0] + res[1]).detach(), res[2][0]._base.detach().clone(), res[2][0]._base._hoc_coverage[0], grad

    assert np.array_equal(np.concatenate([s["hand_verts1"], s["obj_verts1"]], 1), s["verts1"])
    loss_d, flow_d, grad_d = dense()
    assert float(loss_d.abs().sum()) > 0 and float(grad_d.abs().sum()) > 0
    for parts in (False, True):
        loss_f, flow_f, hit, grad_f = fused(parts)
        close(loss_f.cpu().numpy(), loss_d.cpu().numpy(), 2e-6, 1e-9, f"loss (parts: {parts})")
        words = hit.contiguous().view(torch.int32).view(2 * B, (is_ + 7) // 8, (is_ + 31) // 32).cpu().numpy() != 0
        yy, xx = np.mgrid[0:H, 0:Wd]
        defined = torch.from_numpy(words[:, (is_ - 1 - yy) >> 3, xx >> 5]).to(cuda)
        assert torch.equal(flow_f[defined], flow_d[defined]), f"flows (parts: {parts})"
        assert float(flow_d[~defined].abs().sum()) == 0.0  # (the dense flows are exactly zero where nothing is written)
        close(grad_f.cpu().numpy(), grad_d.cpu().numpy(), 1e-5, 1e-6 * float(grad_d.abs().max()), f"vertex gradient (parts: {parts})")


@pytest.mark.parametrize("B,is_,Cj,batched_hand,per_sample_cam", [(3, 256, 3, False, False), (8, 128, 1, True, True), (2, 96, 3, False, False),
                                                                 (5, 480, 1, False, True), (1, 64, 3, False, False)])
def test_pair_step_prologue_in_the_binning_pass(cuda, monkeypatch, B, is_, Cj, batched_hand, per_sample_cam):
    """Round 6: the pair's vertex stage + stacked faces run inside the render's binning pass (bin_boxes_kernel PROLOGUE: the
    image's projected vertices live in LDS only) and the tile-list header is left clean by the previous call's finalize launch
    (MR_PAIR_STEP_LIST_CLEAN) -- against MR_PAIR_STEP_SEPARATE_LAUNCHES, the prologue launch of ABI 8's first form.  The same
    arithmetic on the same inputs: losses, flows under the covered tiles and coverage bytes bit for bit, vertex gradients to the
    order of the scatter's fp32 atomics.  Three calls through ONE plan per form: the first clears the header itself (the
    scratch was never used), the second and third rely on the clean header their predecessor left -- also behind a poisoned
    scratch (0xff everywhere: the host drops LIST_CLEAN) and with a different scene (a stale list would show)."""
    from handobjectconsist_amd.neurender.renderer import Renderer
    from handobjectconsist_amd.warping import opticalflow, pairstep

    ren = Renderer(image_size=is_, R=torch.eye(3, device=cuda)[None], t=torch.zeros(1, 3, device=cuda),
                   K=torch.ones(1, 3, 3, device=cuda), orig_size=is_, anti_aliasing=False, fill_back=True, near=0.1,
                   no_light=True, light_intensity_ambient=0.8)
    if per_sample_cam:
        ang = torch.linspace(-0.05, 0.05, B, device=cuda)
        R = torch.eye(3, device=cuda).repeat(B, 1, 1)
        R[:, 0, 0], R[:, 0, 1], R[:, 1, 0], R[:, 1, 1] = torch.cos(ang), -torch.sin(ang), torch.sin(ang), torch.cos(ang)
        ren.R, ren.t = R, torch.linspace(-0.01, 0.01, B, device=cuda)[:, None].repeat(1, 3).contiguous()
        ren.dist_coeffs = torch.linspace(-0.02, 0.02, B, device=cuda)[:, None] * torch.tensor([1.0, 0.5, 0.1, -0.1, 0.2], device=cuda)
    pairstep._PLANS.clear()  # (plans of earlier tests: theirs have been used, this test starts from a scratch nobody has touched)
    seen = []
    real_fwd = pairstep._PairStepFunction.forward

    def spy(ctx, h1, o1, h2, o2, call):
        out = real_fwd(ctx, h1, o1, h2, o2, call)
        seen.append(int(call[0].st.flags))
        return out

    monkeypatch.setattr(pairstep._PairStepFunction, "forward", staticmethod(spy))
    w = torch.linspace(0.5, 1.5, B, device=cuda)

    def run(flags, seed, poison):
        monkeypatch.setattr(opticalflow, "_PAIR_STEP_FLAGS", flags)
        monkeypatch.setattr(opticalflow, "DEBUG_POISON_RENDER_OUTPUTS", poison)
        s = synth.random_scene(B, seed=seed, image_size=is_)
        im_ref, im, jm_ref, jm = [t(a, cuda) for a in synth.random_images(B, is_, is_, seed + 1)]
        jm_ref, jm = jm_ref[:, :Cj].contiguous(), jm[:, :Cj].contiguous()
        hand_faces = t(s["hand_faces"].astype(np.int64), cuda)
        if batched_hand:
            hand_faces = hand_faces[None].repeat(B, 1, 1)
        obj_faces = t(s["obj_faces"].astype(np.int64)[None].repeat(B, 0), cuda)
        leaves = [t(s[k], cuda).requires_grad_(True) for k in ("hand_verts1", "obj_verts1", "hand_verts2", "obj_verts2")]
        res = opticalflow.flow_pair_loss([(leaves[0], leaves[1]), (leaves[2], leaves[3])], (hand_faces, obj_faces),
                                         [t(s["K1"], cuda), t(s["K2"], cuda)], ren, (is_, is_), im_ref, im, jm_ref, jm,
                                         ignore_face_idxs=synth.HAND_IGNORE_FACES, with_sum=True, with_mean="sum")
        assert res is not None
        lf, lb, flows, lsum, mean = res
        grads = torch.autograd.grad((lf * w).sum() + 0.5 * (lb * w).sum() + 2.0 * mean, leaves)
        base = flows[0]._base
        hit = base._hoc_coverage[0]
        words = hit.contiguous().view(torch.int32).view(2 * B, (is_ + 7) // 8, (is_ + 31) // 32).cpu().numpy() != 0
        yy, xx = np.mgrid[0:is_, 0:is_]
        defined = torch.from_numpy(words[:, (is_ - 1 - yy) >> 3, xx >> 5]).to(cuda)
        return lf.detach(), lb.detach(), mean.detach(), base.detach()[defined], hit.clone(), grads

    calls = ((91, False), (92, True), (93, False))
    got = [run(0, seed, poison) for seed, poison in calls]
    fused_flags = list(seen)
    del seen[:]
    ref = [run(pairstep.SEPARATE_LAUNCHES, seed, poison) for seed, poison in calls]
    assert [f & pairstep.LIST_CLEAN for f in fused_flags] == [0, 0, pairstep.LIST_CLEAN], "first use and poisoned scratch: not clean"
    assert all(f & pairstep.SEPARATE_LAUNCHES for f in seen) and not any(f & pairstep.SEPARATE_LAUNCHES for f in fused_flags)
    for g, r, (seed, _p) in zip(got, ref, calls):
        for x, y, what in zip(g[:5], r[:5], ("loss_fwd", "loss_bwd", "mean", "flows", "coverage bytes")):
            assert torch.equal(x, y), (what, seed)
        assert float(r[0].abs().sum()) > 0
        for x, y, what in zip(g[5], r[5], ("hand 1", "object 1", "hand 2", "object 2")):
            assert torch.isfinite(x).all() and float(y.abs().sum()) > 0, what
            close(x.cpu().numpy(), y.cpu().numpy(), 1e-5, 1e-6 * float(y.abs().max()), f"d/d vertices of {what}, scene {seed}")
