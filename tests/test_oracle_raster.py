"""Known-answer tests pinning the CPU rasteriser oracle (oracle/raster_oracle.c).

Nothing in /root/reference pins the renderer (its arithmetic lives in the absent third-party
`neural_renderer`, and the reference has no tests): these hand-derived cases are the pin
(SURVEY 8c).  Raster orientation: row 0 = bottom of the image, y up."""
import numpy as np
import pytest

from handobjectconsist_amd.utils import synth
from oracle import raster_ref as R

TRI = np.array([[-0.75, -0.75, 1.0], [0.75, -0.75, 2.0], [0.0, 0.75, 4.0]], np.float32)  # CCW in NDC


def tex_for(colors):
    n = len(colors) // 3
    idx = np.arange(3 * n).reshape(1, n, 3)
    return R.batch_vertex_textures(idx, np.asarray(colors, np.float32).reshape(1, 3 * n, 3))


RGB = [[1, 0, 0], [0, 1, 0], [0, 0, 1]]


def render(faces, tex, is_=8, **kw):
    return R.rasterize_forward(np.asarray(faces, np.float32), tex, is_, kw.pop("near", 0.1), kw.pop("far", 100),
                               kw.pop("eps", 1e-3), kw.pop("bg", (0, 0, 0)), **kw)


def test_single_triangle_coverage_and_winding():
    s = render(TRI[None, None], tex_for(RGB))
    expect = np.array([
        [0, 0, 0, 0, 0, 0, 0, 0],
        [0, 1, 1, 1, 1, 1, 1, 0],
        [0, 0, 1, 1, 1, 1, 0, 0],
        [0, 0, 1, 1, 1, 1, 0, 0],
        [0, 0, 0, 1, 1, 0, 0, 0],
        [0, 0, 0, 1, 1, 0, 0, 0],
        [0, 0, 0, 0, 0, 0, 0, 0],
        [0, 0, 0, 0, 0, 0, 0, 0]], bool)
    assert np.array_equal(s["face_index_map"][0] == 0, expect)
    assert np.array_equal(s["alpha_map"][0] == 1, expect)
    assert (s["depth_map"][0][~expect] == 100).all()
    # clockwise copy: back-facing, nothing drawn
    s2 = render(TRI[::-1][None, None], tex_for(RGB))
    assert (s2["face_index_map"] == -1).all() and (s2["alpha_map"] == 0).all()


def test_pixel_centre_on_edge_is_inside():
    # is=4: pixel centres at NDC -0.75, -0.25, 0.25, 0.75.  Edges exactly through centres are inclusive.
    tri = np.array([[-0.75, -0.75, 1], [0.75, -0.75, 1], [-0.75, 0.75, 1]], np.float32)
    s = render(tri[None, None], tex_for(RGB), is_=4)
    expect = np.array([[1, 1, 1, 1], [1, 1, 1, 0], [1, 1, 0, 0], [1, 0, 0, 0]], bool)
    assert np.array_equal(s["face_index_map"][0] >= 0, expect)


def test_tie_goes_to_lowest_index_and_nearest_wins():
    faces = np.stack([TRI, TRI])[None]          # identical faces: equal depth everywhere
    s = render(faces, tex_for(RGB + RGB))
    assert set(np.unique(s["face_index_map"])) == {-1, 0}
    near_tri = TRI * np.array([1, 1, 0.5], np.float32)
    s = render(np.stack([TRI, near_tri])[None], tex_for(RGB + RGB))
    assert set(np.unique(s["face_index_map"])) == {-1, 1}


def test_near_far_rejection():
    for zscale in (0.02, 200.0):  # every vertex closer than near=0.1 / farther than far=100
        s = render((TRI * np.array([1, 1, zscale], np.float32))[None, None], tex_for(RGB))
        assert (s["face_index_map"] == -1).all()
    # a face straddling the near plane is clipped per pixel, not per face
    s = render((TRI * np.array([1, 1, 0.05], np.float32))[None, None], tex_for(RGB))
    hit = s["face_index_map"][0] >= 0
    assert hit.any() and (s["depth_map"][0][hit] > 0.1).all()


def test_weights_and_perspective_depth_closed_form():
    is_ = 64
    s = render(TRI[None, None], tex_for(RGB), is_=is_)
    # pixel (xi, yi) -> NDC centre; barycentrics of the 2-D triangle; 1/z interpolates linearly
    yi, xi = 20, 30
    assert s["face_index_map"][0, yi, xi] == 0
    p = np.array([(2 * xi + 1 - is_) / is_, (2 * yi + 1 - is_) / is_])
    A = np.array([[TRI[0, 0], TRI[1, 0], TRI[2, 0]], [TRI[0, 1], TRI[1, 1], TRI[2, 1]], [1, 1, 1]], np.float64)
    w = np.linalg.solve(A, np.array([p[0], p[1], 1.0]))
    assert np.allclose(s["weight_map"][0, yi, xi], w, atol=1e-5)
    zp = 1.0 / (w / TRI[:, 2].astype(np.float64)).sum()
    assert abs(s["depth_map"][0, yi, xi] - zp) < 1e-5
    # the stored per-pixel inverse maps pixel coordinates to the weights
    inv = s["face_inv_map"][0, yi, xi]
    assert np.allclose(inv @ np.array([xi, yi, 1.0]), w, atol=1e-4)


def test_texture_at_vertex_and_eps_clamp():
    is_ = 64
    eps = 1e-3
    s = render(TRI[None, None], tex_for(RGB), is_=is_, eps=eps)
    # closest pixel to vertex 0: colour ~ (1-eps clamp) * red, weights ~ (1, 0, 0)
    yi, xi = 8, 8
    assert s["face_index_map"][0, yi, xi] == 0
    w, d = s["weight_map"][0, yi, xi].astype(np.float64), float(s["depth_map"][0, yi, xi])
    t = np.clip(w * d / TRI[:, 2], 0, 1 - eps)
    expect = np.array([t[0] * (1 - t[1]) * (1 - t[2]), (1 - t[0]) * t[1] * (1 - t[2]), (1 - t[0]) * (1 - t[1]) * t[2]])
    assert np.allclose(s["rgb_map"][0, yi, xi], expect, atol=1e-5)
    assert s["rgb_map"][0, yi, xi, 0] > 0.8
    # background colour everywhere else
    s = render(TRI[None, None], tex_for(RGB), bg=(0.2, 0.4, 0.6))
    assert np.allclose(s["rgb_map"][0][s["face_index_map"][0] < 0], [0.2, 0.4, 0.6])


def test_flip_orientation_and_fill_back():
    verts = TRI[None]
    K = np.array([[[4.0, 0, 4.0], [0, 4.0, 4.0], [0, 0, 1]]], np.float32)
    cam = (np.eye(3, dtype=np.float32)[None], np.zeros((1, 3), np.float32), np.zeros((1, 5), np.float32), 8, 8)
    cols = np.asarray(RGB, np.float32)[None]

    def go(fidx, fb):
        return R.render(verts, fidx, R.batch_vertex_textures(fidx, cols), K, *cam, fill_back_=fb)

    order = np.array([[[0, 1, 2]]])
    if (go(order, False)["face_index_map"] < 0).all():  # the projection flips y, hence the winding
        order = order[:, :, ::-1]
    vis = go(order, False)
    hidden = order[:, :, ::-1]
    assert (go(hidden, False)["face_index_map"] == -1).all()
    out = go(hidden, True)
    fim = out["face_index_map"][0]
    assert set(np.unique(fim)) == {-1, 1}          # the reversed copy (index F0 + 0) is the visible one
    # image-orientation outputs are the vertical flip of the raster maps; index map stays un-flipped (Q2)
    assert np.array_equal(out["alpha"][0], (fim >= 0).astype(np.float32)[::-1])
    assert out["rgb"].shape == (1, 3, 8, 8) and (fim >= 0).sum() > 4
    assert not np.array_equal(out["alpha"][0], out["alpha"][0][::-1])
    # fill-back permutes the texture so that vertex colours follow the reversed vertex order
    assert np.allclose(out["rgb"], vis["rgb"], atol=1e-6) and np.array_equal(out["alpha"], vis["alpha"])


def test_projection_closed_form():
    K = np.array([[[300.0, 0, 128.0], [0, 310.0, 120.0], [0, 0, 1]]], np.float32)
    v = np.array([[[0.05, -0.02, 0.5], [0.0, 0.0, 0.4]]], np.float32)
    out = R.nr_projection(v, K, np.eye(3, dtype=np.float32)[None], np.zeros((1, 3), np.float32),
                          np.zeros((1, 5), np.float32), 256)
    u = 300 * v[0, :, 0] / v[0, :, 2] + 128
    w = 310 * v[0, :, 1] / v[0, :, 2] + 120
    assert np.allclose(out[0, :, 0], 2 * (u - 128) / 256, atol=1e-5)
    assert np.allclose(out[0, :, 1], 2 * ((256 - w) - 128) / 256, atol=1e-5)
    assert np.allclose(out[0, :, 2], v[0, :, 2])


def scene(B=2, is_=48, seed=0):
    s = synth.random_scene(B, seed=seed, image_size=is_)
    rng = np.random.default_rng(seed)
    tex = R.batch_vertex_textures(s["faces"], rng.uniform(-1, 1, (B, s["verts1"].shape[1], 3)).astype(np.float32))
    f2, t2 = R.fill_back(s["faces"], tex)
    v = R.nr_projection(s["verts1"], s["K1"], np.eye(3, dtype=np.float32)[None], np.zeros((1, 3), np.float32),
                        np.zeros((1, 5), np.float32), is_)
    return R.nr_vertices_to_faces(v, f2), t2


def test_texture_backward_is_the_adjoint_of_sampling():
    faces, tex = scene()
    rng = np.random.default_rng(1)
    s = R.rasterize_forward(faces, tex, 48, 0.1, 100, 1e-3, (0, 0, 0))
    g = rng.standard_normal(s["rgb_map"].shape).astype(np.float32)
    _, gt = R.rasterize_backward(s, g, np.zeros_like(s["alpha_map"]), np.zeros_like(s["depth_map"]))
    hit = s["face_index_map"] >= 0
    lhs = float((s["rgb_map"].astype(np.float64) * g)[hit].sum())        # <C t, g>  (background = 0)
    rhs = float((tex.astype(np.float64) * gt).sum())                     # <t, E g>
    assert abs(lhs - rhs) < 1e-4 * max(abs(lhs), 1.0)
    assert hit.sum() > 100


def test_depth_backward_matches_finite_differences():
    is_ = 32
    tri = np.array([[-0.8, -0.7, 1.0], [0.9, -0.6, 1.5], [0.1, 0.85, 2.5]], np.float32)[None, None]
    s = render(tri, tex_for(RGB), is_=is_)
    yi, xi = 12, 15
    assert s["face_index_map"][0, yi, xi] == 0
    g = np.zeros_like(s["depth_map"]); g[0, yi, xi] = 1.0
    gf = R.backward_depth_map(tri, s["depth_map"], s["face_index_map"], s["face_inv_map"], s["weight_map"], g,
                              np.zeros_like(tri))
    for k in range(3):
        for c in range(3):
            h = 1e-3
            plus, minus = tri.copy(), tri.copy()
            plus[0, 0, k, c] += h; minus[0, 0, k, c] -= h
            fd = (render(plus, tex_for(RGB), is_=is_)["depth_map"][0, yi, xi].astype(np.float64)
                  - render(minus, tex_for(RGB), is_=is_)["depth_map"][0, yi, xi]) / (2 * h)
            assert abs(gf[0, 0, k, c] - fd) < 2e-2 * max(1.0, abs(fd)), (k, c, gf[0, 0, k, c], fd)


def test_pixel_map_pseudo_gradient_sign_and_support():
    is_ = 32
    tri = (TRI * np.array([0.6, 0.6, 1.0], np.float32))[None, None]
    s = render(tri, tex_for([[1, 1, 1]] * 3), is_=is_)
    zeros = np.zeros_like(s["alpha_map"])
    gf0, _ = R.rasterize_backward(s, np.zeros_like(s["rgb_map"]), zeros, np.zeros_like(s["depth_map"]))
    assert np.abs(gf0).max() == 0
    # L = -sum(alpha): descending L grows the silhouette, i.e. -grad points away from the centroid
    gf, _ = R.rasterize_backward(s, np.zeros_like(s["rgb_map"]), -np.ones_like(zeros), np.zeros_like(s["depth_map"]))
    assert np.isfinite(gf).all() and np.abs(gf[0, 0, :, :2]).max() > 0 and np.abs(gf[0, 0, :, 2]).max() == 0
    centroid = tri[0, 0, :, :2].mean(0)
    for k in range(3):
        assert np.dot(-gf[0, 0, k, :2], tri[0, 0, k, :2] - centroid) > 0
    # back-facing faces receive nothing
    both = np.concatenate([tri, tri[:, :, ::-1]], 1)
    s2 = render(both, tex_for([[1, 1, 1]] * 6), is_=is_)
    gf2, _ = R.rasterize_backward(s2, np.zeros_like(s2["rgb_map"]), -np.ones_like(zeros), np.zeros_like(s2["depth_map"]))
    assert np.abs(gf2[0, 1]).max() == 0 and np.allclose(gf2[0, 0], gf[0, 0])


def test_config1_hand_silhouette_antialiased():
    """BASELINE config 1: single 1538-face hand-like mesh, 64x64 silhouette, AA on (128^2 raster)."""
    s = synth.random_scene(1, seed=3, image_size=64)
    fidx = s["hand_faces"][None, :1538]
    f2, _ = R.fill_back(fidx)
    v = R.nr_projection(s["hand_verts1"], s["K1"], np.eye(3, dtype=np.float32)[None], np.zeros((1, 3), np.float32),
                        np.zeros((1, 5), np.float32), 64)
    out = R.rasterize_rgbad(R.nr_vertices_to_faces(v, f2), None, 64, True, return_rgb=False, return_alpha=True,
                            return_depth=False)
    a = out["alpha"]
    assert a.shape == (1, 64, 64) and out["face_index_map"].shape == (1, 128, 128) and out["rgb"] is None
    assert set(np.unique(a)).issubset({0.0, 0.25, 0.5, 0.75, 1.0}) and 0 < a.mean() < 0.5
    assert ((a > 0) & (a < 1)).sum() > 0  # anti-aliased boundary pixels
