"""The oracle's restatement of the reference's Python glue (oracle/raster_ref.py: RasterizeFunction,
rasterize_rgbad, Renderer.render; oracle/warp_ref.py: get_opticalflow, pair_consist) against fixtures
produced by RUNNING that glue (tests/golden/make_golden_chain.py: /root/reference's rasterize.py,
renderer.py, opticalflow.py, imgflowarp.py, warpbranch.py executed on CPU, the absent third-party
kernels stubbed by the C oracle).  CPU only.

The six kernels themselves stay "parity unpinned" (third-party source absent): both sides of these
comparisons call the same C restatement for them.  What is pinned here is every line between the
kernels: buffer pre-fills, alpha / background, NHWC->NCHW + vertical flip (and which maps are not
flipped), anti-aliasing, fill-back, the mask algebra of get_opticalflow, crop, GT-reference
substitution, detach of frames > 0 and stack().mean().
"""
import json
import os

import numpy as np
import pytest

from oracle import raster_ref as R
from oracle import warp_ref as W

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name))
    return z, json.loads(str(z["meta"]))


def relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def test_rasterize_rgbad_matches_reference_glue():
    z, meta = load("chain_rasterize.npz")
    assert len(meta) >= 26
    for m in meta:
        k = m["key"]
        tex = z[f"textures_ts{m['ts']}"] if m["return_rgb"] else None
        out = R.rasterize_rgbad(z["faces"], tex, m["image_size"], m["anti_aliasing"], m["near"], m["far"], m["eps"],
                                m["background_value"], m["return_rgb"], m["return_alpha"], m["return_depth"],
                                keep_saved=True)
        for name, want in (("rgb", m["return_rgb"]), ("alpha", m["return_alpha"]), ("depth", m["return_depth"])):
            if want:
                np.testing.assert_array_equal(out[name], z[f"{k}_{name}"], err_msg=f"{k} {name}")
            else:
                assert out[name] is None and f"{k}_{name}" not in z.files
        np.testing.assert_array_equal(out["face_index_map"], z[f"{k}_face_index_map"])
        np.testing.assert_array_equal(out["weight_map"], z[f"{k}_weight_map"])
        np.testing.assert_array_equal(out["face_inv_map"], z[f"{k}_face_inv_map"])
        # backward: adjoint of avg-pool + flip + permute, then RasterizeFunction.backward
        s = m["image_size"]

        def up(g, chw):
            if g is None:
                return None
            if m["anti_aliasing"]:
                g = np.repeat(np.repeat(g, 2, axis=-1), 2, axis=-2) * np.float32(0.25)
            g = g[..., ::-1, :]
            return np.ascontiguousarray(g.transpose(0, 2, 3, 1) if chw else g, np.float32)

        gf, gt = R.rasterize_backward(
            out["_saved"], up(z[f"g_rgb_{s}"], True) if m["return_rgb"] else None,
            up(z[f"g_alpha_{s}"], False) if m["return_alpha"] else None,
            up(z[f"g_depth_{s}"], False) if m["return_depth"] else None)
        assert relerr(gf, z[f"{k}_grad_faces"]) < 1e-6, k
        if m["return_rgb"]:
            assert relerr(gt, z[f"{k}_grad_textures"]) < 1e-6, k
    np.testing.assert_array_equal(R.rasterize_rgbad(z["faces"], z["textures_ts2"], 12, True, return_alpha=False,
                                                    return_depth=False)["rgb"], z["w_rasterize"])
    np.testing.assert_array_equal(R.rasterize_rgbad(z["faces"], None, 12, True, return_rgb=False,
                                                    return_depth=False)["alpha"], z["w_silhouettes"])
    np.testing.assert_array_equal(R.rasterize_rgbad(z["faces"], None, 12, False, return_rgb=False,
                                                    return_alpha=False)["depth"], z["w_depth"])


def _ctor(z, ctor):
    return {k: (z[v] if isinstance(v, str) and v in z.files else v) for k, v in ctor.items()}


def test_renderer_render_matches_reference_glue():
    z, meta = load("chain_renderer.npz")
    seen = 0
    for m in meta:
        if m["kind"] != "render":
            continue
        c = _ctor(z, m["ctor"])
        call = _ctor(z, m["call"])
        K = call.get("K", c.get("K"))
        tex = z["textures"]
        faces_idx = z["faces"]
        if not c.get("no_light", False):
            # lighting acts on the fill-backed faces in WORLD coordinates (renderer.py:254-265)
            f2, t2 = R.fill_back(faces_idx, tex) if c["fill_back"] else (faces_idx, tex)
            lit = R.nr_lighting(R.nr_vertices_to_faces(z["verts"], f2), t2,
                                c.get("light_intensity_ambient", 0.5), c.get("light_intensity_directional", 0.5),
                                c.get("light_color_ambient", (1, 1, 1)), c.get("light_color_directional", (1, 1, 1)),
                                c.get("light_direction", (0, 1, 0)))
            v = R.nr_projection(z["verts"], K, c["R"], c["t"], c.get("dist_coeffs", np.zeros((1, 5), np.float32)),
                                c["orig_size"])
            out = R.rasterize_rgbad(R.nr_vertices_to_faces(v, f2), lit, c["image_size"], c["anti_aliasing"],
                                    c.get("near", 0.1), c.get("far", 100), 1e-3, c.get("background_color", (0, 0, 0)))
        else:
            out = R.render(z["verts"], faces_idx, tex, K, c["R"], c["t"],
                           c.get("dist_coeffs", np.zeros((1, 5), np.float32)), c["orig_size"], c["image_size"],
                           c["anti_aliasing"], c["fill_back"], c.get("near", 0.1), c.get("far", 100), 1e-3,
                           c.get("background_color", (0, 0, 0)))
        k = m["key"]
        # projection runs through numpy matmul here and torch matmul in the fixture: allow a rounding
        # difference in the projected vertices to move a handful of edge pixels
        mism = int((out["face_index_map"] != z[f"{k}_face_index_map"]).sum())
        assert mism <= 2, (k, mism)
        same = (out["face_index_map"] == z[f"{k}_face_index_map"])
        if c["anti_aliasing"]:
            same = same.reshape(same.shape[0], same.shape[1] // 2, 2, same.shape[2] // 2, 2).all(axis=(2, 4))
        same_img = same[:, ::-1]
        for name in ("rgb", "alpha", "depth"):
            a, b = out[name], z[f"{k}_{name}"]
            sel = same_img[:, None] if a.ndim == 4 else same_img
            assert np.abs((a - b) * sel).max() <= 2e-4 * max(1.0, np.abs(b).max()), (k, name)
        seen += 1
    assert seen >= 7


@pytest.mark.parametrize("scene", ["sq", "crop", "one"])
def test_get_opticalflow_matches_reference_glue(scene):
    z, meta = load("chain_opticalflow.npz")
    ran = 0
    for m in meta:
        if m["scene"] != scene or not m["detach_renders"]:
            continue  # (values do not depend on the detach flags; the attached case is a gradient fixture)
        is_ = m["image_size"]
        kw = dict(R=np.eye(3, dtype=np.float32)[None], t=np.zeros((1, 3), np.float32),
                  dist_coeffs=np.zeros((1, 5), np.float32), orig_size=is_, image_size=is_, anti_aliasing=False,
                  near=0.1, far=100, eps=1e-3)
        flows = W.get_opticalflow(R, [z[f"{scene}_verts1"], z[f"{scene}_verts2"]], z[f"{scene}_faces"],
                                  [z[f"{scene}_K1"], z[f"{scene}_K2"]], kw,
                                  orig_img_size=m["orig_img_size"], mask_occlusions=m["mask_occlusions"],
                                  ignore_face_idxs=m["ignore_face_idxs"] if m["ignore"] else None)
        for i, name in enumerate(("flow12", "flow21")):
            want = z[f"{m['key']}_{name}"]
            assert flows[i].shape == want.shape
            support = int(((flows[i] != 0) != (want != 0)).sum())
            assert support <= 4, (m["key"], name, support)
            both = (flows[i] != 0) & (want != 0)
            assert np.abs((flows[i] - want) * both).max() < 5e-3, (m["key"], name)
            assert np.median(np.abs(flows[i] - want)[both]) < 1e-5
        ran += 1
    assert ran >= 1


def test_get_opticalflow_config_size_matches_reference_glue():
    """The oracle chain at the raster size of BASELINE.json's config 2 (480, crop 480 x 270) against the flows the
    reference's own get_opticalflow returned (40 000 seeded pixels per flow + support counts)."""
    z, meta = load("chain_opticalflow_cfg.npz")
    m = next(mm for mm in meta if mm["scene"] == "c480")
    s, is_ = m["scene"], m["image_size"]
    kw = dict(R=np.eye(3, dtype=np.float32)[None], t=np.zeros((1, 3), np.float32),
              dist_coeffs=np.zeros((1, 5), np.float32), orig_size=is_, image_size=is_, anti_aliasing=False,
              near=0.1, far=100, eps=1e-3)
    flows = W.get_opticalflow(R, [z[f"{s}_verts1"], z[f"{s}_verts2"]], z[f"{s}_faces"], [z[f"{s}_K1"], z[f"{s}_K2"]], kw,
                              orig_img_size=tuple(m["orig_img_size"]), mask_occlusions=True,
                              ignore_face_idxs=m["ignore_face_idxs"])
    idx = z[f"{s}_sample_idx"]
    for i, name in enumerate(("flow12", "flow21")):
        got = flows[i].reshape(-1, 2)
        want = z[f"{s}_{name}_sample"]
        # (numpy matmul here, torch matmul in the fixture: a rounding difference in the projected vertices may move
        # a handful of edge pixels)
        assert abs(int((got[:, 0] != 0).sum()) - int(z[f"{s}_{name}_support"][0])) <= 4
        both = (got[idx] != 0) & (want != 0)
        assert int(((got[idx] != 0) != (want != 0)).sum()) <= 4 and both.sum() > 1000
        assert np.abs((got[idx] - want) * both).max() < 5e-3
        assert np.median(np.abs(got[idx] - want)[both]) < 1e-5


def test_warpbranch_matches_reference_glue():
    """warpbranch.forward (warpbranch.py:28-96) restated with the oracle: which vertices feed frame k,
    flows per (0, k) pair, pair_consist per pair, mean over pairs."""
    z, meta = load("chain_warpbranch.npz")
    for m in meta:
        is_, crop, k = m["image_size"], m["input_res"], m["key"]
        kw = dict(R=np.eye(3, dtype=np.float32)[None], t=np.zeros((1, 3), np.float32),
                  dist_coeffs=np.zeros((1, 5), np.float32), orig_size=is_, image_size=is_, anti_aliasing=False,
                  near=0.1, far=100, eps=1e-3)
        verts = []
        for f in range(m["frames"]):
            src = "gt" if (m["gt_refs"] and f > 0) else "pred"
            verts.append(np.concatenate([z[f"f{f}_{src}_hand"], z[f"f{f}_{src}_obj"]], 1))
        B = verts[0].shape[0]
        Vh = z["f0_pred_hand"].shape[1]
        faces = np.concatenate([z["hand_face"][None].repeat(B, 0), z[f"f{m['frames'] - 1}_objfaces"] + Vh], 1)
        losses = []
        for p in range(1, m["frames"]):
            flows = W.get_opticalflow(R, [verts[0], verts[p]], faces, [z["f0_camintr"], z[f"f{p}_camintr"]], kw,
                                      orig_img_size=crop, ignore_face_idxs=m["hand_ignore_faces"])
            for d in (0, 1):
                want = z[f"{k}_p{p - 1}_flow{d}"]
                assert int(((flows[d] != 0) != (want != 0)).sum()) <= 4
            # loss from the REFERENCE's flows: isolates pair_consist + the mean from render rounding
            ref_flows = [z[f"{k}_p{p - 1}_flow0"], z[f"{k}_p{p - 1}_flow1"]]
            loss, masks, warps, diffs, _ = W.pair_consist(ref_flows, z["f0_image"], z[f"f{p}_image"],
                                                       z["f0_jittermask"], z[f"f{p}_jittermask"], m["use_backward"])
            for d in (0, 1):
                np.testing.assert_array_equal(masks[d]["full_mask"], z[f"{k}_p{p - 1}_full_mask{d}"])
                assert np.abs(warps[d] - z[f"{k}_p{p - 1}_warp{d}"]).max() < 2e-6
            losses.append(loss)
        losses = np.stack(losses)
        assert relerr(losses, z[f"{k}_diff_losses"]) < 1e-5
        assert abs(losses.mean() - float(z[f"{k}_loss"])) < 1e-5 * abs(float(z[f"{k}_loss"]))
