"""The HIP path, through the package's drop-in Python surface, against fixtures produced by RUNNING THE
REFERENCE'S OWN GLUE (tests/golden/make_golden_chain.py executes /root/reference's rasterize.py,
renderer.py, opticalflow.py, imgflowarp.py, pyramidloss.py and warpbranch.py on CPU; the absent
third-party kernels are stubbed by the C oracle -- those six kernels stay "parity unpinned").

Compared: every output of rasterize_rgbad / Renderer.* / get_opticalflow / warpbranch.forward AND the
gradients the reference's autograd chain delivers at the mesh vertices / textures / faces.
Tolerances: face_index_map exact, images 1e-6 when both sides rasterise the SAME projected faces
(chain_rasterize), gradients 1e-4 relative to the gradient's norm (north-star tolerance).  Where the
vertices are projected on both sides independently (torch CPU matmul in the fixture, HIP vertex stage /
rocBLAS here) a 1-ulp difference in a projected vertex may move an edge pixel: a handful of support
mismatches is allowed and values are compared on the common support (README "projected-vertex caveat").
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name))
    return z, json.loads(str(z["meta"]))


def t(a, dev, grad=False):
    x = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return x.requires_grad_(True) if grad else x


def n(x):
    return x.detach().cpu().numpy()


def norm_rel(a, b):
    """max |a - b| relative to max |b| (norm-relative: gradients span many orders of magnitude)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def l2_rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


# ---------------------------------------------------------------------------------------------------


@pytest.mark.parametrize("fused", [True, False])
def test_rasterize_rgbad_against_reference_glue(cuda, fused):
    """26 configurations of rasterize_rgbad (AA, every return_* combination, tuple / [B,3] background,
    ts 2 and 3, both eps values) + RasterizeFunction.backward, fused kernels and the 5-entry-point path."""
    from handobjectconsist_amd.neurender import rasterize

    z, meta = load("chain_rasterize.npz")
    rasterize.USE_FUSED = fused
    try:
        for m in meta:
            k, s = m["key"], m["image_size"]
            f = t(z["faces"], cuda, True)
            x = t(z[f"textures_ts{m['ts']}"], cuda, True) if m["return_rgb"] else None
            out = rasterize.rasterize_rgbad(f, x, s, m["anti_aliasing"], m["near"], m["far"], m["eps"],
                                            m["background_value"], m["return_rgb"], m["return_alpha"],
                                            m["return_depth"])
            assert np.array_equal(n(out["face_index_map"]), z[f"{k}_face_index_map"]), k
            outs, grads = [], []
            for name, want in (("rgb", m["return_rgb"]), ("alpha", m["return_alpha"]), ("depth", m["return_depth"])):
                if not want:
                    assert out[name] is None, (k, name)
                    continue
                ref = z[f"{k}_{name}"]
                assert out[name].shape == ref.shape
                assert np.abs(n(out[name]) - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()), (k, name)
                outs.append(out[name])
                grads.append(t(z[f"g_{name}_{s}"], cuda))
            hit = z[f"{k}_face_index_map"] >= 0
            assert np.abs((n(out["weight_map"]) - z[f"{k}_weight_map"]) * hit[..., None]).max() <= 1e-6, k
            if m["return_depth"]:
                fi, ref = n(out["face_inv_map"]), z[f"{k}_face_inv_map"]
                assert fi.shape == ref.shape
                assert np.abs((fi - ref) * hit[..., None, None]).max() <= 1e-6 * max(1.0, np.abs(ref).max()), k
            torch.autograd.backward(outs, grads)
            gf = z[f"{k}_grad_faces"]
            assert norm_rel(n(f.grad), gf) < 1e-4, (k, "grad_faces", norm_rel(n(f.grad), gf))
            if m["return_rgb"]:
                gt = z[f"{k}_grad_textures"]
                assert norm_rel(n(x.grad), gt) < 1e-4, (k, "grad_textures")
        f = t(z["faces"], cuda)
        assert np.abs(n(rasterize.rasterize(f, t(z["textures_ts2"], cuda), 12)) - z["w_rasterize"]).max() <= 1e-6
        assert np.abs(n(rasterize.rasterize_silhouettes(f, 12)) - z["w_silhouettes"]).max() <= 1e-6
        assert np.abs(n(rasterize.rasterize_depth(f, 12, False)) - z["w_depth"]).max() <= 1e-6
    finally:
        rasterize.USE_FUSED = True


# ---------------------------------------------------------------------------------------------------


def _resolve(z, d, dev):
    return {k: (t(z[v], dev) if isinstance(v, str) and v in z.files else v) for k, v in d.items()}


def _common_support(fim, ref_fim, aa):
    """Pixels (IMAGE orientation, output resolution) whose winning face agrees on both sides."""
    same = fim == ref_fim
    if aa:
        same = same.reshape(same.shape[0], same.shape[1] // 2, 2, same.shape[2] // 2, 2).all(axis=(2, 4))
    return same[:, ::-1]


def test_renderer_against_reference_glue(cuda):
    """Renderer.render as WarpRegNet / fastrender build it (per-sample K at construction and per call,
    rotation + translation + lens distortion, fill-back off, anti-aliasing + background, lighting),
    values and the gradients at vertices and textures."""
    from handobjectconsist_amd.neurender.renderer import Renderer

    z, meta = load("chain_renderer.npz")
    ran = 0
    for m in meta:
        if m["kind"] != "render":
            continue
        k = m["key"]
        ren = Renderer(**_resolve(z, m["ctor"], cuda))
        v, x = t(z["verts"], cuda, True), t(z["textures"], cuda, True)
        out = ren(v, t(z["faces"], cuda), x, detach_renders=m["detach_renders"], **_resolve(z, m["call"], cuda))
        aa = m["ctor"]["anti_aliasing"]
        mism = int((n(out["face_index_map"]) != z[f"{k}_face_index_map"]).sum())
        assert mism <= 2, (k, mism)
        same = _common_support(n(out["face_index_map"]), z[f"{k}_face_index_map"], aa)
        for name in ("rgb", "alpha", "depth"):
            a, b = n(out[name]), z[f"{k}_{name}"]
            sel = same[:, None] if a.ndim == 4 else same
            assert np.abs((a - b) * sel).max() <= 2e-4 * max(1.0, np.abs(b).max()), (k, name)
        (out["rgb"] * t(z["g_rgb"], cuda)).sum().add((out["alpha"] * t(z["g_alpha"], cuda)).sum()).add(
            (out["depth"] * t(z["g_depth"], cuda)).sum()).backward()
        if mism == 0:
            assert norm_rel(n(x.grad), z[f"{k}_grad_textures"]) < 1e-4, (k, "grad_textures")
            if m["detach_renders"]:
                # the reference's autograd left vertices.grad = None: nothing may arrive here either
                assert f"{k}_grad_verts" not in z.files
                assert v.grad is None or float(v.grad.abs().max()) == 0.0
            else:
                gv = z[f"{k}_grad_verts"]
                # the pseudo-gradient of kernel D has 1/distance terms: norm-relative
                assert norm_rel(n(v.grad), gv) < 2e-4, (k, "grad_verts", norm_rel(n(v.grad), gv))
        ran += 1
    assert ran >= 7


def test_renderer_on_the_references_projected_vertices(cuda):
    """Why test_renderer_against_reference_glue allows 2 support mismatches and 2e-4: the fixture projects the vertices
    with torch's CPU matmul, this package with rocBLAS / the HIP vertex stage, and a 1-ulp difference in a projected
    vertex can move an edge pixel.  Shown, not assumed: with ``Renderer.project`` handing out the projected vertices the
    REFERENCE fed its rasteriser (``*_ndc`` of the fixture), every configuration reproduces face_index_map exactly,
    the images to 1e-6, and the gradients at the textures AND at the projected vertices to the north-star 1e-4."""
    from handobjectconsist_amd.neurender.renderer import Renderer

    z, meta = load("chain_renderer.npz")
    ran = 0
    for m in meta:
        if m["kind"] != "render":
            continue
        k = m["key"]
        ren = Renderer(**_resolve(z, m["ctor"], cuda))
        ndc = t(z[f"{k}_ndc"], cuda, True)
        ren.project = lambda vertices, *a, **kw: ndc  # (instance attribute: shadows the method for this renderer only)
        v, x = t(z["verts"], cuda, True), t(z["textures"], cuda, True)
        out = ren(v, t(z["faces"], cuda), x, detach_renders=m["detach_renders"], **_resolve(z, m["call"], cuda))
        assert np.array_equal(n(out["face_index_map"]), z[f"{k}_face_index_map"]), (k, "face_index_map")
        for name in ("rgb", "alpha", "depth", "weight_map"):
            a, b = n(out[name]), z[f"{k}_{name}"]
            assert np.abs(a - b).max() <= 1e-6 * max(1.0, np.abs(b).max()), (k, name, np.abs(a - b).max())
        (out["rgb"] * t(z["g_rgb"], cuda)).sum().add((out["alpha"] * t(z["g_alpha"], cuda)).sum()).add(
            (out["depth"] * t(z["g_depth"], cuda)).sum()).backward()
        assert norm_rel(n(x.grad), z[f"{k}_grad_textures"]) < 1e-4, (k, "grad_textures", norm_rel(n(x.grad), z[f"{k}_grad_textures"]))
        if m["detach_renders"]:
            assert f"{k}_grad_ndc" not in z.files and ndc.grad is None
        else:
            want = z[f"{k}_grad_ndc"]
            assert np.abs(want).max() > 0
            assert norm_rel(n(ndc.grad), want) < 1e-4, (k, "grad_ndc", norm_rel(n(ndc.grad), want))
        ran += 1
    assert ran >= 7


def test_renderer_modes_against_reference_glue(cuda):
    """mode='rgb' / 'silhouettes' / 'depth' (module-default eps and near / far), project, and the
    look_at / look cameras."""
    from handobjectconsist_amd.neurender.renderer import Renderer

    z, meta = load("chain_renderer.npz")
    ran = 0
    for m in meta:
        k = m["key"]
        if m["kind"] in ("rgb", "silhouettes", "depth", "project"):
            ren = Renderer(**_resolve(z, m["ctor"], cuda))
            v, x = t(z["verts"], cuda, True), t(z["textures"], cuda, True)
            if m["kind"] == "project":
                out = ren.project(v)
                g = torch.ones_like(out) * torch.tensor([1.0, -2.0, 0.5], device=cuda)
                assert np.abs(n(out) - z[f"{k}_out"]).max() < 2e-6 * max(1.0, np.abs(z[f"{k}_out"]).max())
            else:
                out = ren(v, t(z["faces"], cuda), x, mode=m["kind"])
                g = t(z["g_rgb"] if m["kind"] == "rgb" else z["g_alpha"], cuda)
                ref = z[f"{k}_out"]
                assert out.shape == ref.shape
                bad = np.abs(n(out) - ref) > 2e-4 * max(1.0, np.abs(ref).max())
                assert int(bad.sum()) <= 2 * (3 if out.dim() == 4 else 1), (k, int(bad.sum()))
            (out * g).sum().backward()
            if m["kind"] == "project":
                assert norm_rel(n(v.grad), z[f"{k}_grad_verts"]) < 1e-4
            elif m["kind"] == "rgb":
                assert l2_rel(n(x.grad), z[f"{k}_grad_textures"]) < 2e-2  # (an edge pixel may differ)
            ran += 1
        elif m["kind"] == "render_unit":
            ren = Renderer(**m["ctor"])
            v, x = t(z["verts_unit"], cuda, True), t(z["textures"], cuda, True)
            out = ren(v, t(z["faces"], cuda), x)
            aa = m["ctor"]["anti_aliasing"]
            mism = int((n(out["face_index_map"]) != z[f"{k}_face_index_map"]).sum())
            assert mism <= 2, (k, mism)
            same = _common_support(n(out["face_index_map"]), z[f"{k}_face_index_map"], aa)
            for name in ("rgb", "alpha", "depth"):
                a, b = n(out[name]), z[f"{k}_{name}"]
                sel = same[:, None] if a.ndim == 4 else same
                assert np.abs((a - b) * sel).max() <= 2e-4 * max(1.0, np.abs(b).max()), (k, name)
            ran += 1
    assert ran >= 8


# ---------------------------------------------------------------------------------------------------


def _training_renderer(is_, dev):
    from handobjectconsist_amd.neurender.renderer import Renderer

    return Renderer(image_size=is_, R=torch.eye(3, device=dev).unsqueeze(0), t=torch.zeros(1, 3, device=dev),
                    K=torch.ones(1, 3, 3, device=dev), orig_size=is_, anti_aliasing=False, fill_back=True, near=0.1,
                    no_light=True)


def _flow_check(got, want, key, max_support=4):
    support = int(((got != 0) != (want != 0)).sum())
    assert support <= max_support, (key, "support", support)
    both = (got != 0) & (want != 0)
    assert both.sum() > 20, key
    assert np.abs((got - want) * both).max() < 5e-3, (key, np.abs((got - want) * both).max())
    assert np.median(np.abs(got - want)[both]) < 1e-5, key
    return support


@pytest.mark.parametrize("path", ["fused", "fused_poisoned", "fused_dense_tiles", "fused_separate_nodes", "fused_full_outputs",
                                  "vertex_color", "textures"])
def test_get_opticalflow_against_reference_glue(cuda, path):
    """Flows and d(flows)/d(vertices of both frames) for every fixture variant (ignore list, crop,
    detach_textures, detach_renders=False, mask_occlusions=False), through the fused vertex stage +
    stacked render (what training launches), the per-frame vertex-colour render, and the materialised
    texture path (the reference's own op sequence on the HIP rasteriser)."""
    from handobjectconsist_amd.warping import opticalflow

    z, meta = load("chain_opticalflow.npz")
    saved = (opticalflow.USE_FUSED_VERTEX_STAGE, opticalflow.USE_VERTEX_COLOR_RENDER, opticalflow.USE_FLOW_RENDER,
             opticalflow.USE_STACKED_FLOW_NODE, opticalflow.USE_SPARSE_TILES, opticalflow.DEBUG_POISON_RENDER_OUTPUTS)
    opticalflow.USE_FUSED_VERTEX_STAGE = path.startswith("fused")
    opticalflow.USE_VERTEX_COLOR_RENDER = path != "textures"
    one_node = path in ("fused", "fused_poisoned", "fused_dense_tiles")
    opticalflow.USE_FLOW_RENDER = one_node or path == "fused_separate_nodes"  # flow-mode render (mask folded in)
    opticalflow.USE_STACKED_FLOW_NODE = one_node  # one autograd node, single fused backward launch
    opticalflow.USE_SPARSE_TILES = path != "fused_dense_tiles"  # the render skips tiles without candidate faces
    # pixels the sparse render leaves unwritten hold NaN / INT_MIN: nothing downstream may have read them
    opticalflow.DEBUG_POISON_RENDER_OUTPUTS = path == "fused_poisoned"
    try:
        for m in meta:
            s, k, is_ = m["scene"], m["key"], m["image_size"]
            v1, v2 = t(z[f"{s}_verts1"], cuda, True), t(z[f"{s}_verts2"], cuda, True)
            flows = opticalflow.get_opticalflow(
                [v1, v2], t(z[f"{s}_faces"], cuda), [t(z[f"{s}_K1"], cuda), t(z[f"{s}_K2"], cuda)],
                _training_renderer(is_, cuda), orig_img_size=m["orig_img_size"], mask_occlusions=m["mask_occlusions"],
                detach_textures=m["detach_textures"], detach_renders=m["detach_renders"],
                ignore_face_idxs=m["ignore_face_idxs"] if m["ignore"] else None)
            sup = 0
            for i, name in enumerate(("flow12", "flow21")):
                assert tuple(flows[i].shape) == z[f"{k}_{name}"].shape
                sup += _flow_check(n(flows[i]), z[f"{k}_{name}"], (path, k, name))
            if path.startswith("fused"):
                # the vertex stage reproduces the fixture's projections bit for bit: nothing may differ in support,
                # so the strict (north-star) gradient tolerance below is the one that is exercised
                assert sup == 0, (k, sup)
            ((flows[0] * t(z[f"{s}_g12"], cuda)).sum() + (flows[1] * t(z[f"{s}_g21"], cuda)).sum()).backward()
            for v, name in ((v1, "grad_verts1"), (v2, "grad_verts2")):
                want = z[f"{k}_{name}"]
                got = n(v.grad) if v.grad is not None else np.zeros_like(want)
                if sup == 0 and m["detach_renders"]:
                    # exact adjoint chain (pair of renders -> textures -> projections): north-star tolerance
                    assert norm_rel(got, want) < 1e-4, (path, k, name, norm_rel(got, want))
                elif sup == 0:
                    assert norm_rel(got, want) < 1e-3, (path, k, name, norm_rel(got, want))  # + kernel D's 1/dist terms
                else:
                    assert l2_rel(got, want) < 5e-2, (path, k, name, l2_rel(got, want))
    finally:
        (opticalflow.USE_FUSED_VERTEX_STAGE, opticalflow.USE_VERTEX_COLOR_RENDER, opticalflow.USE_FLOW_RENDER,
         opticalflow.USE_STACKED_FLOW_NODE, opticalflow.USE_SPARSE_TILES, opticalflow.DEBUG_POISON_RENDER_OUTPUTS) = saved


@pytest.mark.parametrize("poisoned", [False, True])
def test_get_opticalflow_at_baseline_config_sizes_against_reference_glue(cuda, poisoned):
    """The training path (stacked node: flow-mode render with sparse tiles and per-pixel records, occlusion + epilogue,
    one backward launch) at the raster sizes of BASELINE.json's configs -- 480 (crop 480 x 270) and 640 (crop
    640 x 480) -- against what the reference's own get_opticalflow returned on CPU: flows at 40 000 seeded pixels, their
    support counts and sums, and d loss / d vertices of both frames in full (north-star tolerance)."""
    from handobjectconsist_amd.warping import opticalflow

    z, meta = load("chain_opticalflow_cfg.npz")
    saved = opticalflow.DEBUG_POISON_RENDER_OUTPUTS
    opticalflow.DEBUG_POISON_RENDER_OUTPUTS = poisoned
    try:
        for m in meta:
            s, is_, (W, H), B = m["scene"], m["image_size"], m["orig_img_size"], m["batch"]
            v1, v2 = t(z[f"{s}_verts1"], cuda, True), t(z[f"{s}_verts2"], cuda, True)
            flows = opticalflow.get_opticalflow(
                [v1, v2], t(z[f"{s}_faces"], cuda), [t(z[f"{s}_K1"], cuda), t(z[f"{s}_K2"], cuda)],
                _training_renderer(is_, cuda), orig_img_size=(W, H), mask_occlusions=True, detach_textures=False,
                detach_renders=True, ignore_face_idxs=m["ignore_face_idxs"])
            assert hasattr(flows[0]._base, "_hoc_coverage"), "the stacked training node must have been taken"
            idx = z[f"{s}_sample_idx"]
            for i, name in enumerate(("flow12", "flow21")):
                assert tuple(flows[i].shape) == (B, H, W, 2)
                got = n(flows[i]).reshape(-1, 2)
                want = z[f"{s}_{name}_sample"]
                assert np.array_equal((got[:, 0] != 0).sum(), z[f"{s}_{name}_support"][0]), (s, name, "support")
                assert np.array_equal(got[idx] != 0, want != 0), (s, name, "sampled support")
                assert np.abs(got[idx] - want).max() <= 1e-6 * max(np.abs(want).max(), 1.0), (s, name)
                assert np.allclose(got.astype(np.float64).sum(0), z[f"{s}_{name}_sum"], rtol=1e-5, atol=1e-3), (s, name, "sum")
            r = np.random.default_rng(m["grad_seed"])  # = tests/golden/make_golden_chain.py::flow_grad_inputs
            g12 = r.standard_normal((B, H, W, 2)).astype(np.float32)
            g21 = r.standard_normal((B, H, W, 2)).astype(np.float32)
            ((flows[0] * t(g12, cuda)).sum() + (flows[1] * t(g21, cuda)).sum()).backward()
            for v, name in ((v1, "grad_verts1"), (v2, "grad_verts2")):
                want = z[f"{s}_{name}"]
                assert np.abs(want).max() > 0
                assert norm_rel(n(v.grad), want) < 1e-4, (s, name, norm_rel(n(v.grad), want))
    finally:
        opticalflow.DEBUG_POISON_RENDER_OUTPUTS = saved


def test_flow_render_equals_full_render(cuda):
    """mr_render_flow_forward (the training path's output set) against mr_render_vc_forward + mr_flow_mask on the
    same inputs: displacement planes, alpha, mask and face_index_map bit-equal; the colour gradient (barycentrics
    recomputed instead of read back from the weight / depth maps) bit-equal too."""
    from handobjectconsist_amd.utils import synth
    from handobjectconsist_amd.warping import opticalflow

    for B, is_, seed in ((3, 96, 3), (2, 256, 4)):
        s = synth.random_scene(B, seed=seed, image_size=is_)
        ren = _training_renderer(is_, cuda)
        ndc = ren.project(t(s["verts1"], cuda), K=t(s["K1"], cuda)).detach()
        faces = t(s["faces"], cuda)
        rng = np.random.default_rng(seed)
        g = t(rng.standard_normal((B, 3, is_, is_)).astype(np.float32), cuda)
        g[:, 2] = 0  # the third plane of the flow render is never written (and never read downstream)
        ignore = list(range(100, 900)) + list(range(3552 + 200, 3552 + 700))  # visible faces of both orientations
        lut = opticalflow._keep_lut(ignore, cuda)
        c1 = t(rng.standard_normal((B, ndc.shape[1], 3)).astype(np.float32), cuda, True)
        full = ren.render_projected_vertex_colors(ndc, faces, c1)
        m_full, _ = opticalflow._flow_mask(full, ignore)
        (full["rgb"] * g).sum().backward()
        c2 = c1.detach().clone().requires_grad_(True)
        flow = ren.render_projected_flow(ndc, faces, c2, lut)
        (flow["rgb"][:, :2] * g[:, :2]).sum().backward()
        assert torch.equal(flow["rgb"][:, :2], full["rgb"][:, :2])
        assert torch.equal(flow["alpha"], full["alpha"]) and torch.equal(flow["face_index_map"], full["face_index_map"])
        assert torch.equal(flow["mask"], m_full) and float(m_full.sum()) < float(full["alpha"].sum())
        assert norm_rel(n(c2.grad), n(c1.grad)) < 1e-6


def test_flow_finalize_backward_against_reference_glue(cuda):
    """mr_flow_finalize_backward in isolation: the adjoint of ``(rgb * mask_pre * (mask_x * occl))
    .permute(0,2,3,1)[..., :2][:, :H, :W]`` (opticalflow.py:118,146-154).  The masks are recovered from the
    reference's flows, the incoming gradient is the fixture's, and the result must be what torch autograd gives
    for the reference's own op sequence."""
    from handobjectconsist_amd import _lib

    z, meta = load("chain_opticalflow.npz")
    m = next(mm for mm in meta if mm["scene"] == "crop")
    is_, (W, H) = m["image_size"], m["orig_img_size"]
    B = z["crop_g12"].shape[0]
    rng = np.random.default_rng(0)
    rgb = t(rng.standard_normal((B, 3, is_, is_)).astype(np.float32), cuda, True)
    mask_pre = t((rng.random((B, is_, is_)) < 0.7).astype(np.float32), cuda)
    mask_x = t(rng.random((B, is_, is_)).astype(np.float32), cuda)  # raw alpha may be any float (Q4)
    occl = t((rng.random((B, is_, is_)) < 0.8).astype(np.float32), cuda)
    g = t(z["crop_g12"], cuda)
    ref = ((rgb * mask_pre.unsqueeze(1)) * (mask_x * occl).unsqueeze(1)).permute(0, 2, 3, 1)[:, :, :, :2][:, :H, :W]
    (ref * g).sum().backward()
    flow = torch.empty((B, H, W, 2), dtype=torch.float32, device=cuda)
    _lib.call("mr_flow_finalize_forward", _lib.ptr(rgb.detach()), _lib.ptr(mask_pre), _lib.ptr(mask_x), _lib.ptr(occl),
              _lib.ptr(flow), B, is_, H, W, _lib.stream_ptr(cuda))
    assert np.abs(n(flow) - n(ref)).max() <= 1e-6
    grad_rgb = torch.full((B, 3, is_, is_), float("nan"), dtype=torch.float32, device=cuda)
    _lib.call("mr_flow_finalize_backward", _lib.ptr(g), _lib.ptr(mask_pre), _lib.ptr(mask_x), _lib.ptr(occl),
              _lib.ptr(grad_rgb), B, is_, H, W, _lib.stream_ptr(cuda))
    assert torch.isfinite(grad_rgb).all(), "backward must write every element (also outside the crop and channel 2)"
    assert np.abs(n(grad_rgb) - n(rgb.grad)).max() <= 1e-6 * float(rgb.grad.abs().max())


# ---------------------------------------------------------------------------------------------------


@pytest.mark.parametrize("mode", ["full", "loss"])
@pytest.mark.parametrize("keys", ["enum", "string"])
def test_warpbranch_forward_against_reference_glue(cuda, keys, mode, monkeypatch):
    """warpbranch.forward (warpbranch.py:9-96): GT-reference substitution, detach of frames > 0
    (first_only), per-pair pair_consist, stack().mean(), and d loss / d predicted vertices of every frame.
    ``mode`` "loss" = the trainer's setting (pair_outputs="loss": fused pair nodes, flows defined under their renders only,
    no per-pixel outputs): same losses and gradients as the reference's run."""
    from handobjectconsist_amd.datasets.queries import BaseQueries as BQ
    from handobjectconsist_amd.datasets.queries import TransQueries as TQ
    from handobjectconsist_amd.models import warpbranch
    from handobjectconsist_amd.optim.pyramidloss import PyramidCriterion
    from handobjectconsist_amd.warping import opticalflow

    # "loss": render outputs start as NaN, so whatever the kernels leave unwritten shows (and dense_flows must mask it)
    monkeypatch.setattr(opticalflow, "DEBUG_POISON_RENDER_OUTPUTS", mode == "loss")
    z, meta = load("chain_warpbranch.npz")
    names = {"image": TQ.IMAGE, "jittermask": TQ.JITTERMASK, "camintr": TQ.CAMINTR, "objfaces": BQ.OBJFACES,
             "objverts3d": BQ.OBJVERTS3D, "handverts3d": BQ.HANDVERTS3D}
    key = (lambda s: names[s]) if keys == "enum" else (lambda s: s)
    for m in meta:
        k, is_, crop = m["key"], m["image_size"], tuple(m["input_res"])
        samples, results = [], []
        for f in range(m["frames"]):
            samples.append({key("image"): t(z[f"f{f}_image"], cuda), key("jittermask"): t(z[f"f{f}_jittermask"], cuda),
                            key("camintr"): t(z[f"f{f}_camintr"], cuda), key("objfaces"): t(z[f"f{f}_objfaces"], cuda),
                            key("objverts3d"): t(z[f"f{f}_gt_obj"], cuda),
                            key("handverts3d"): t(z[f"f{f}_gt_hand"], cuda)})
            results.append({"recov_handverts3d": t(z[f"f{f}_pred_hand"], cuda, True),
                            "recov_objverts3d": t(z[f"f{f}_pred_obj"], cuda, True)})
        loss, pair = warpbranch.forward(
            samples, results, t(z["hand_face"], cuda)[None], _training_renderer(is_, cuda), crop,
            PyramidCriterion("l1"), gt_refs=m["gt_refs"], first_only=m["first_only"],
            hand_ignore_faces=m["hand_ignore_faces"], use_backward=m["use_backward"], pair_outputs=mode)
        loss.backward()
        sup = 0
        for p in range(m["frames"] - 1):
            for d in (0, 1):
                if mode == "loss":  # flows: wherever the reference's are non-zero (unspecified memory elsewhere)
                    got, want = n(pair["recons_flows"][p][d]), z[f"{k}_p{p}_flow{d}"]
                    on = want[..., 0] != 0
                    assert on.sum() > 20 and np.abs(got[on] - want[on]).max() < 5e-3, (k, p, d, np.abs(got[on] - want[on]).max())
                    assert np.median(np.abs(got[on] - want[on])) < 1e-5, (k, p, d)
                    # ... and on request as the reference returns them: zeros wherever nothing was rendered
                    dense = n(opticalflow.dense_flows(pair["recons_flows"][p])[d])
                    assert _flow_check(dense, want, (k, p, d)) == 0, (k, p, d, "support of the zero-filled flows")
                    continue
                sup += _flow_check(n(pair["recons_flows"][p][d]), z[f"{k}_p{p}_flow{d}"], (k, p, d))
                assert sup == 0, (k, p, d, "the training path must reproduce the support of the reference's flows")
                fm = n(pair["masks"][p][d]["full_mask"]).astype(bool)
                assert int((fm != z[f"{k}_p{p}_full_mask{d}"]).sum()) <= 4, (k, p, d)
                wm = n(pair["masks"][p][d]["warp_mask"])[:, 0]
                assert int((wm != z[f"{k}_p{p}_warp_mask{d}"]).sum()) <= 4, (k, p, d)
                both = fm & z[f"{k}_p{p}_full_mask{d}"]
                assert np.abs((n(pair["warps"][p][d]) - z[f"{k}_p{p}_warp{d}"]) * both[:, None]).max() < 1e-4
        tol = 1e-4 if sup == 0 else 2e-2
        assert norm_rel(n(pair["diff_losses"]), z[f"{k}_diff_losses"]) < tol, (k, "diff_losses")
        assert abs(float(loss) - float(z[f"{k}_loss"])) < tol * abs(float(z[f"{k}_loss"])), (k, "loss")
        for f, res in enumerate(results):
            for name in ("recov_handverts3d", "recov_objverts3d"):
                want = z[f"{k}_f{f}_grad_{name}"]
                got = n(res[name].grad) if res[name].grad is not None else np.zeros_like(want)
                if np.abs(want).max() == 0:
                    assert np.abs(got).max() == 0, (k, f, name, "frame must not receive a gradient")
                elif sup == 0:
                    assert norm_rel(got, want) < 1e-4, (k, f, name, norm_rel(got, want))
                else:
                    assert l2_rel(got, want) < 5e-2, (k, f, name, l2_rel(got, want))


def test_fastrender_render_against_the_reference(cuda):
    """meshreg/neurender/fastrender.py:14-59 run from the reference (tests/golden/make_golden_chain.py fastrender): lit RGBA
    render, crop / no crop, background compositing.  The reference's compositing line multiplies [B,3,H,W] by [B,H,W],
    which only broadcasts for B = 1 (the fixture records that B = 2 raises there; B = 3 would silently take the alpha of
    sample c for channel c): the mirror composites per sample -- a conscious fix, equal to the reference where it works."""
    from handobjectconsist_amd.neurender import fastrender

    z, meta = load("chain_fastrender.npz")
    ran = 0
    for m in meta:
        k = m["key"]
        if k == "b2_bg_raises_in_the_reference":
            assert m["value"] is True
            continue
        out = fastrender.render(t(z[f"{k}_verts"], cuda), t(z[f"{k}_faces"], cuda), tuple(m["input_res"]), camintrs=t(z[f"{k}_K"], cuda),
                                colors=t(z[f"{k}_colors"], cuda), bg_color=m["bg_color"], crop_to_img=m["crop_to_img"])
        want = z[f"{k}_out"]
        assert tuple(out.shape) == want.shape, (k, out.shape, want.shape)
        got = n(out)
        support = (got[..., 3] > 0) != (want[..., 3] > 0)
        assert int(support.sum()) <= 2, (k, int(support.sum()))  # (independent projections: README caveat)
        assert np.abs((got - want) * ~support[..., None]).max() <= 2e-4, (k, np.abs((got - want) * ~support[..., None]).max())
        ran += 1
    assert ran == 3
    # the per-sample compositing where the reference cannot run: alpha of sample b, for every channel of sample b
    k = "b2"
    m = [x for x in meta if x["key"] == k][0]
    plain = fastrender.render(t(z[f"{k}_verts"], cuda), t(z[f"{k}_faces"], cuda), tuple(m["input_res"]), camintrs=t(z[f"{k}_K"], cuda),
                              colors=t(z[f"{k}_colors"], cuda))
    comp = fastrender.render(t(z[f"{k}_verts"], cuda), t(z[f"{k}_faces"], cuda), tuple(m["input_res"]), camintrs=t(z[f"{k}_K"], cuda),
                             colors=t(z[f"{k}_colors"], cuda), bg_color=0.5)
    a = plain[..., 3:4]
    assert torch.allclose(comp[..., :3], plain[..., :3] * a + 0.5 * (1 - a), atol=1e-6) and torch.equal(comp[..., 3], plain[..., 3])
