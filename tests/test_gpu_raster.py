"""Parity of the HIP rasteriser (through the C-ABI) against the CPU oracle.

face_index_map must match exactly; weight / depth / rgb are bit-identical by construction
(same fp32 operation order, -ffp-contract=off on both sides) and are checked to 1e-6;
gradients to 1e-4 relative (the north-star tolerance) where the summation order differs.
"""
import numpy as np
import pytest
import torch

from handobjectconsist_amd.utils import synth
from oracle import raster_ref as R

pytestmark = pytest.mark.gpu

REN_KW = dict(R=np.eye(3, dtype=np.float32)[None], t=np.zeros((1, 3), np.float32),
              dist_coeffs=np.zeros((1, 5), np.float32), near=0.1, far=100, eps=1e-3)


def projected_faces(B, image_size, seed, fill_back=True):
    """NDC faces + vertex-colour textures of the synthetic hand+object scene."""
    s = synth.random_scene(B, seed=seed, image_size=image_size)
    rng = np.random.default_rng(seed)
    colors = rng.uniform(-2, 2, (B, s["verts1"].shape[1], 3)).astype(np.float32)
    tex = R.batch_vertex_textures(s["faces"], colors)
    fidx, tex = R.fill_back(s["faces"], tex) if fill_back else (s["faces"], tex)
    v = R.nr_projection(s["verts1"], s["K1"], REN_KW["R"], REN_KW["t"], REN_KW["dist_coeffs"], image_size)
    return R.nr_vertices_to_faces(v, fidx), tex


def big_faces(B, image_size, seed, n=24):
    """A handful of large random triangles (exercise the wave-cooperative paths), both windings."""
    rng = np.random.default_rng(seed)
    f = rng.uniform(-1.3, 1.3, (B, n, 3, 3)).astype(np.float32)
    f[..., 2] = rng.uniform(0.3, 3.0, (B, n, 3))
    f = np.concatenate([f, f[:, :, ::-1]], 1)
    tex = rng.uniform(-1, 1, (B, 2 * n, 2, 2, 2, 3)).astype(np.float32)
    return np.ascontiguousarray(f), tex


def t(a, dev):
    # (a copy, not ascontiguousarray: a reversed axis of length 1 counts as contiguous and keeps its negative stride)
    return torch.from_numpy(np.array(a, copy=True, order="C")).to(dev)


def assert_close(a, b, rtol, atol, what):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    assert (err <= tol).all(), f"{what}: max err {err.max():.3e} (|ref| max {np.abs(b).max():.3e}), {(err > tol).sum()} / {err.size} out of tol"


CASES = [("scene", 2, 64, 0), ("scene", 3, 96, 1), ("scene", 2, 256, 2), ("big", 2, 64, 3), ("big", 1, 100, 4)]


def make_case(kind, B, is_, seed):
    return projected_faces(B, is_, seed) if kind == "scene" else big_faces(B, is_, seed)


@pytest.mark.parametrize("kind,B,is_,seed", CASES)
@pytest.mark.parametrize("reference_algo", [False, True])
def test_fused_forward_matches_oracle(cuda, kind, B, is_, seed, reference_algo):
    from handobjectconsist_amd.neurender import rasterize

    if reference_algo and is_ > 100:
        pytest.skip("brute-force validation kernel only at small sizes")
    faces, tex = make_case(kind, B, is_, seed)
    bg = (0.25, -0.5, 0.75)
    ref = R.rasterize_rgbad(faces, tex, is_, False, 0.1, 100, 1e-3, bg, num_threads=8)
    rasterize.REFERENCE_ALGO = reference_algo
    try:
        out = rasterize.rasterize_rgbad(t(faces, cuda), t(tex, cuda), is_, False, 0.1, 100, 1e-3, bg)
    finally:
        rasterize.REFERENCE_ALGO = False
    fim = out["face_index_map"].cpu().numpy()
    assert fim.dtype == np.int32
    mism = (fim != ref["face_index_map"]).sum()
    assert mism == 0, f"face_index_map differs at {mism} pixels"
    assert (fim >= 0).sum() > 50
    assert_close(out["weight_map"].cpu().numpy(), ref["weight_map"], 0, 1e-6, "weight_map")
    assert_close(out["depth"].cpu().numpy(), ref["depth"], 1e-6, 0, "depth")
    assert_close(out["alpha"].cpu().numpy(), ref["alpha"], 0, 0, "alpha")
    assert_close(out["rgb"].cpu().numpy(), ref["rgb"], 1e-6, 1e-6, "rgb")
    assert_close(out["face_inv_map"].cpu().numpy(), ref["face_inv_map"], 1e-6, 1e-7, "face_inv_map")
    assert set(out.keys()) == {"rgb", "alpha", "depth", "face_inv_map", "face_index_map", "weight_map"}


def test_raster_beyond_the_bin_counters(cuda):
    """A 1536-pixel raster has more tiles (9216) than the binning pass has LDS counters (8192): a bin is then a column of
    two tiles, and a bin's records may miss one of its tiles -- they get an empty row range in the tile kernel, as do
    the records of the image's large list (faces over more than 8 bins) in the tiles they do not touch."""
    from handobjectconsist_amd.neurender import rasterize

    is_, rng = 1536, np.random.default_rng(11)
    n_small = 120
    c = rng.uniform(-0.97, 0.97, (1, n_small, 1, 2))
    small = np.concatenate([c + rng.uniform(-0.02, 0.02, (1, n_small, 3, 2)), rng.uniform(0.5, 2.0, (1, n_small, 3, 1))], -1)
    big, _ = big_faces(1, is_, 12, n=4)
    faces = np.concatenate([small, small[:, :, ::-1], big], 1).astype(np.float32)
    tex = rng.uniform(-1, 1, (1, faces.shape[1], 2, 2, 2, 3)).astype(np.float32)
    bg = (0.1, 0.2, 0.3)
    ref = R.rasterize_rgbad(faces, tex, is_, False, 0.1, 100, 1e-3, bg, num_threads=8)
    out = rasterize.rasterize_rgbad(t(faces, cuda), t(tex, cuda), is_, False, 0.1, 100, 1e-3, bg)
    fim = out["face_index_map"].cpu().numpy()
    assert (fim != ref["face_index_map"]).sum() == 0
    won = np.unique(fim[fim >= 0])
    assert (won < 2 * n_small).sum() > 60 and (won >= 2 * n_small).sum() >= 2, "small and large faces must both be visible"
    assert_close(out["weight_map"].cpu().numpy(), ref["weight_map"], 0, 1e-6, "weight_map")
    assert_close(out["depth"].cpu().numpy(), ref["depth"], 1e-6, 0, "depth")
    assert_close(out["rgb"].cpu().numpy(), ref["rgb"], 1e-6, 1e-6, "rgb")


@pytest.mark.parametrize("kind,B,is_,seed", CASES[:2] + CASES[3:4])
def test_compat_five_entry_points_match_oracle(cuda, kind, B, is_, seed):
    """RasterizeFunction = the reference's structure on the 5 upstream-compatible entry points."""
    from handobjectconsist_amd.neurender import rasterize

    faces, tex = make_case(kind, B, is_, seed)
    bg = (0.1, 0.2, 0.3)
    saved = R.rasterize_forward(faces, tex, is_, 0.1, 100, 1e-3, bg, num_threads=8)
    f_t = t(faces, cuda).requires_grad_(True)
    x_t = t(tex, cuda).requires_grad_(True)
    rgb, alpha, depth, fim, finv, wmap = rasterize.RasterizeFunction.apply(
        f_t, x_t, is_, 0.1, 100, 1e-3, bg, True, True, True)
    assert (fim.cpu().numpy() != saved["face_index_map"]).sum() == 0
    assert_close(rgb.detach().cpu().numpy(), saved["rgb_map"], 1e-6, 1e-6, "rgb_map")
    assert_close(alpha.detach().cpu().numpy(), saved["alpha_map"], 0, 0, "alpha_map")
    assert_close(depth.detach().cpu().numpy(), saved["depth_map"], 1e-6, 0, "depth_map")
    assert_close(wmap.detach().cpu().numpy(), saved["weight_map"], 0, 1e-6, "weight_map")
    assert_close(finv.detach().cpu().numpy(), saved["face_inv_map"], 1e-6, 1e-7, "face_inv_map")
    rng = np.random.default_rng(seed + 100)
    g_rgb = rng.standard_normal(saved["rgb_map"].shape).astype(np.float32)
    g_alpha = rng.standard_normal(saved["alpha_map"].shape).astype(np.float32)
    g_depth = rng.standard_normal(saved["depth_map"].shape).astype(np.float32)
    gf_ref, gt_ref = R.rasterize_backward(saved, g_rgb, g_alpha, g_depth, num_threads=8)
    torch.autograd.backward([rgb, alpha, depth], [t(g_rgb, cuda), t(g_alpha, cuda), t(g_depth, cuda)])
    scale_t = np.abs(gt_ref).max()
    assert_close(x_t.grad.cpu().numpy(), gt_ref, 1e-4, 1e-5 * scale_t, "grad_textures")
    scale_f = np.abs(gf_ref).max()
    assert_close(f_t.grad.cpu().numpy(), gf_ref, 1e-4, 1e-5 * scale_f, "grad_faces")


def _img_grads(saved, seed):
    rng = np.random.default_rng(seed + 100)
    g_rgb = rng.standard_normal(saved["rgb_map"].shape).astype(np.float32)       # raster NHWC
    g_alpha = rng.standard_normal(saved["alpha_map"].shape).astype(np.float32)
    g_depth = rng.standard_normal(saved["depth_map"].shape).astype(np.float32)
    img = (np.ascontiguousarray(g_rgb.transpose(0, 3, 1, 2)[:, :, ::-1]), np.ascontiguousarray(g_alpha[:, ::-1]),
           np.ascontiguousarray(g_depth[:, ::-1]))
    return (g_rgb, g_alpha, g_depth), img


@pytest.mark.parametrize("kind,B,is_,seed", CASES)
@pytest.mark.parametrize("reference_algo", [False, True])
def test_fused_backward_matches_oracle(cuda, kind, B, is_, seed, reference_algo):
    from handobjectconsist_amd.neurender import rasterize

    faces, tex = make_case(kind, B, is_, seed)
    bg = (0.0, 0.0, 0.0)
    ref = R.rasterize_rgbad(faces, tex, is_, False, 0.1, 100, 1e-3, bg, num_threads=8, keep_saved=True)
    saved = ref["_saved"]
    raster_g, img_g = _img_grads(saved, seed)
    gf_ref, gt_ref = R.rasterize_backward(saved, *raster_g, num_threads=8)
    f_t = t(faces, cuda).requires_grad_(True)
    x_t = t(tex, cuda).requires_grad_(True)
    rasterize.REFERENCE_ALGO = reference_algo
    try:
        out = rasterize.rasterize_rgbad(f_t, x_t, is_, False, 0.1, 100, 1e-3, bg)
        torch.autograd.backward([out["rgb"], out["alpha"], out["depth"]], [t(g, cuda) for g in img_g])
    finally:
        rasterize.REFERENCE_ALGO = False
    scale_t = np.abs(gt_ref).max()
    assert_close(x_t.grad.cpu().numpy(), gt_ref, 1e-4, 1e-5 * scale_t, "grad_textures")
    scale_f = np.abs(gf_ref).max()
    assert_close(f_t.grad.cpu().numpy(), gf_ref, 1e-4, 1e-5 * scale_f, "grad_faces")


@pytest.mark.parametrize("kind,B,is_,seed", [("big", 1, 67, 11), ("scene", 9, 64, 12), ("scene", 1, 320, 13), ("big", 2, 300, 14),
                                            ("scene", 1, 600, 15), ("big", 1, 1100, 16), ("big", 1, 1500, 17)])
def test_fused_backward_strip_widths(cuda, kind, B, is_, seed):
    """Kernel D by strips (raster_bwd.hip) beyond the sizes of CASES: an odd raster (partial last strip, flags marked pixel
    by pixel), more images than XCDs, rasters that take two lines (257..527) and one line (528..1421) per strip, and one
    too wide for LDS (1500: the plane-reading walk) -- all against the C oracle's ordered walk."""
    from handobjectconsist_amd.neurender import rasterize

    faces, tex = make_case(kind, B, is_, seed)
    bg = (0.0, 0.0, 0.0)
    ref = R.rasterize_rgbad(faces, tex, is_, False, 0.1, 100, 1e-3, bg, num_threads=8, keep_saved=True)
    saved = ref["_saved"]
    raster_g, img_g = _img_grads(saved, seed)
    gf_ref, gt_ref = R.rasterize_backward(saved, *raster_g, num_threads=8)
    f_t = t(faces, cuda).requires_grad_(True)
    x_t = t(tex, cuda).requires_grad_(True)
    out = rasterize.rasterize_rgbad(f_t, x_t, is_, False, 0.1, 100, 1e-3, bg)
    torch.autograd.backward([out["rgb"], out["alpha"], out["depth"]], [t(g, cuda) for g in img_g])
    assert np.abs(gf_ref).max() > 0
    # (sweeps of a thousand and more terms: the order of the fp32 additions -- sequential in the oracle, by lanes on the
    # GPU -- shows at 5e-5 of the largest gradient on the 1500-pixel raster)
    floor = 1e-5 if is_ < 1000 else 1e-4
    assert_close(x_t.grad.cpu().numpy(), gt_ref, 1e-4, floor * np.abs(gt_ref).max(), "grad_textures")
    assert_close(f_t.grad.cpu().numpy(), gf_ref, 1e-4, floor * np.abs(gf_ref).max(), "grad_faces")


def test_strip_chunks_behind_a_sweeps_end_stay_finite(cuda):
    """Fuzz seed 91002: an "out" sweep that runs towards pixel 0 ends next to the edge; the positions of its last chunk
    BEHIND its end (weighted 0) cross the edge, where the distance k * (d1 - d1_cross) +- eps passes through exactly 0 --
    0 * (1 / 0) must not reach the sums.  One face with pixel-lattice vertices, eps = 0.01, raster 57."""
    from handobjectconsist_amd.neurender import rasterize

    tri = np.array([[[-0.75, -1.25, 2.5], [0.875, 1.0, 0.625], [-0.25, 0.0, 0.125]]], np.float32)
    faces = np.ascontiguousarray(np.stack([tri[0], tri[0][::-1]])[None])
    tex = np.random.default_rng(5).uniform(-1, 1, (1, 2, 2, 2, 2, 3)).astype(np.float32)
    ref = R.rasterize_rgbad(faces, tex, 57, False, 0.1, 100, 1e-2, (0.1, 0.2, 0.3), num_threads=2, keep_saved=True)
    saved = ref["_saved"]
    raster_g, img_g = _img_grads(saved, 91002)
    gf_ref, gt_ref = R.rasterize_backward(saved, *raster_g, num_threads=2)
    f_t, x_t = t(faces, cuda).requires_grad_(True), t(tex, cuda).requires_grad_(True)
    out = rasterize.rasterize_rgbad(f_t, x_t, 57, False, 0.1, 100, 1e-2, (0.1, 0.2, 0.3))
    torch.autograd.backward([out["rgb"], out["alpha"], out["depth"]], [t(g, cuda) for g in img_g])
    assert np.isfinite(gf_ref).all() and torch.isfinite(f_t.grad).all()
    assert_close(f_t.grad.cpu().numpy(), gf_ref, 1e-4, 1e-5 * np.abs(gf_ref).max(), "grad_faces")


@pytest.mark.parametrize("n", [1, 3, 7])
def test_fused_backward_few_faces(cuda, n):
    """2n faces of one image: most strips of kernel D have no owner to walk, the others one or two."""
    from handobjectconsist_amd.neurender import rasterize

    faces, tex = big_faces(1, 64, 40 + n, n=n)
    ref = R.rasterize_rgbad(faces, tex, 64, False, 0.1, 100, 1e-3, (0.0, 0.0, 0.0), num_threads=8, keep_saved=True)
    saved = ref["_saved"]
    raster_g, img_g = _img_grads(saved, n)
    gf_ref, gt_ref = R.rasterize_backward(saved, *raster_g, num_threads=8)
    f_t, x_t = t(faces, cuda).requires_grad_(True), t(tex, cuda).requires_grad_(True)
    out = rasterize.rasterize_rgbad(f_t, x_t, 64, False, 0.1, 100, 1e-3, (0.0, 0.0, 0.0))
    torch.autograd.backward([out["rgb"], out["alpha"], out["depth"]], [t(g, cuda) for g in img_g])
    assert np.abs(gf_ref).max() > 0
    assert_close(x_t.grad.cpu().numpy(), gt_ref, 1e-4, 1e-5 * np.abs(gt_ref).max(), "grad_textures")
    assert_close(f_t.grad.cpu().numpy(), gf_ref, 1e-4, 1e-5 * np.abs(gf_ref).max(), "grad_faces")


@pytest.mark.parametrize("is_", [1, 2, 3, 5, 8])
def test_fused_backward_on_rasters_of_a_few_pixels(cuda, is_):
    """Kernel D's strip bookkeeping (2 B is / L weight words, cleared by the marking pass) has more words than the image has
    pixel quads at 1 x 1 and 2 x 2 pixels: the call refused such rasters ("bad argument"; found by tests/fuzz_parity.py)."""
    from handobjectconsist_amd.neurender import rasterize

    faces, tex = big_faces(3, is_, 70 + is_, n=6)
    ref = R.rasterize_rgbad(faces, tex, is_, False, 0.1, 100, 1e-3, (0.1, 0.2, 0.3), num_threads=8, keep_saved=True)
    saved = ref["_saved"]
    raster_g, img_g = _img_grads(saved, is_)
    gf_ref, gt_ref = R.rasterize_backward(saved, *raster_g, num_threads=8)
    f_t, x_t = t(faces, cuda).requires_grad_(True), t(tex, cuda).requires_grad_(True)
    out = rasterize.rasterize_rgbad(f_t, x_t, is_, False, 0.1, 100, 1e-3, (0.1, 0.2, 0.3))
    assert (out["face_index_map"].cpu().numpy() == ref["face_index_map"]).all()
    torch.autograd.backward([out["rgb"], out["alpha"], out["depth"]], [t(g, cuda) for g in img_g])
    assert_close(x_t.grad.cpu().numpy(), gt_ref, 1e-4, 1e-5 * max(np.abs(gt_ref).max(), 1e-30), "grad_textures")
    assert_close(f_t.grad.cpu().numpy(), gf_ref, 2e-4, 1e-5 * max(np.abs(gf_ref).max(), 1e-30), "grad_faces")


@pytest.mark.parametrize("B,is_,sparse", [(2, 64, False), (3, 96, True), (2, 256, True)])
def test_flow_pixel_records_equal_stored_maps(cuda, B, is_, sparse):
    """mr_render_flow_forward with a vertex-id map leaves per-pixel records (the winner's vertex ids + the sampling
    weights of its three colour taps) instead of barycentrics + depth; mr_render_flow_backward on the records gives the
    gradient of the stored-map path (same products, same fixed-point sums per workgroup), for both gradient forms."""
    from handobjectconsist_amd import _lib

    d = _vc_abi_case(cuda, B, is_, 21)
    P, st = _lib.ptr, _lib.stream_ptr(cuda)
    f32 = dict(dtype=torch.float32, device=cuda)
    V, F0 = d["V"], d["F0"]
    bg = torch.zeros(3, **f32)
    wbytes = int(_lib.load().mr_render_workspace_bytes(B, 2 * F0, is_))
    work = torch.empty((wbytes,), dtype=torch.uint8, device=cuda)
    outs = {}
    for rec in (False, True):
        rgb, alpha, mask = torch.empty((B, 3, is_, is_), **f32), torch.empty((B, is_, is_), **f32), torch.empty((B, is_, is_), **f32)
        depth = torch.full((B, is_, is_), float("nan"), **f32)
        wmap = torch.full((B, is_, is_, 3), float("nan"), **f32)
        fim = torch.full((B, is_, is_), -7, dtype=torch.int32, device=cuda)
        vid = torch.full((B, is_, is_, 3), -7, dtype=torch.int32, device=cuda)
        hit = torch.empty((B, (is_ + 7) // 8, (is_ + 31) // 32, 4), dtype=torch.uint8, device=cuda)
        _lib.call("mr_render_flow_forward", P(d["v"]), P(d["fidx"]), P(d["cols"]), P(bg), 0, None, 0, 0.99999, P(rgb), P(alpha),
                  P(mask), None if rec else P(depth), P(wmap), P(fim), P(hit), P(work), wbytes, B, V, F0, 1, is_, 0.1, 100.0,
                  1e-3, _lib.FLAG_SPARSE_TILES if sparse else 0, P(vid) if rec else None, 0, None, None, 0, 0, st)
        g = torch.Generator(device="cpu").manual_seed(5)
        g_rgb = torch.randn((B, 3, is_, is_), generator=g).to(cuda)
        gf = torch.randn((B, is_, is_, 2), generator=g).to(cuda)
        m_pre = (torch.rand((B, is_, is_), generator=g) < 0.8).float().to(cuda)
        m_x = torch.rand((B, is_, is_), generator=g).to(cuda)
        occl = (torch.rand((B, is_, is_), generator=g) < 0.9).float().to(cuda)
        grads = []
        for flowgrad in (False, True):
            out = torch.full((B, V, 3), float("nan"), **f32)
            _lib.call("mr_render_flow_backward", None if rec else P(d["v"]), None if rec else P(d["fidx"]), P(fim), P(hit), P(wmap),
                      None if rec else P(depth), None if flowgrad else P(g_rgb), P(gf), P(m_pre), P(m_x), None, B, P(occl), is_, is_,
                      P(out), B, V, F0, 1, is_, 1e-3, 0, P(vid) if rec else None, 0, None, st)
            grads.append(out)
        outs[rec] = (rgb, fim, hit, grads, vid, wmap)
    covered = outs[True][1] >= 0
    if not sparse:
        assert torch.equal(outs[False][1], outs[True][1])
    assert torch.equal(outs[False][2], outs[True][2]), "coverage bytes"
    assert int(covered.sum()) > 100
    vid, wrec = outs[True][4], outs[True][5]
    assert int((vid[covered] < 0).sum()) == 0 and int((vid[covered] >= V).sum()) == 0
    assert torch.isfinite(wrec[covered]).all() and float(wrec[covered].min()) >= 0 and float(wrec[covered].max()) <= 1
    for a, b_ in zip(outs[False][3], outs[True][3]):
        assert torch.isfinite(a).all() and float(a.abs().sum()) > 0
        # (per workgroup the sums are exact fixed-point integers of identical products; the 16 workgroups of an image
        # then meet in fp32 global atomics, whose order varies from run to run)
        assert_close(b_.cpu().numpy(), a.cpu().numpy(), 1e-5, 1e-6 * float(a.abs().max()), "gradient on the pixel records")


@pytest.mark.parametrize("B,is_", [(3, 96), (2, 256), (5, 64)])
def test_flow_forward_tile_list_equals_one_workgroup_per_tile(cuda, B, is_):
    """Sparse-tile launches over the compacted list of tiles with candidates (tile_bound != 0): whatever the caller's
    guess of the list length -- 1 (almost everything goes through the looping overflow launch), exact, far too large,
    'auto' -- every output byte equals the one-workgroup-per-tile launch, on poisoned buffers (nothing else is
    written), and the reported list length is the number of tiles with candidates."""
    from handobjectconsist_amd import _lib

    d = _vc_abi_case(cuda, B, is_, 33)
    P, st = _lib.ptr, _lib.stream_ptr(cuda)
    f32 = dict(dtype=torch.float32, device=cuda)
    V, F0 = d["V"], d["F0"]
    bg = torch.zeros(3, **f32)
    wbytes = int(_lib.load().mr_render_workspace_bytes(B, 2 * F0, is_))
    work = torch.empty((wbytes,), dtype=torch.uint8, device=cuda)
    word = torch.zeros(1, dtype=torch.int32).pin_memory()

    def run(bound, flags=0):
        rgb = torch.full((B, 3, is_, is_), float("nan"), **f32)
        alpha, mask = torch.full((B, is_, is_), float("nan"), **f32), torch.full((B, is_, is_), float("nan"), **f32)
        wmap = torch.full((B, is_, is_, 3), float("nan"), **f32)
        fim = torch.full((B, is_, is_), -7, dtype=torch.int32, device=cuda)
        vid = torch.full((B, is_, is_, 3), -7, dtype=torch.int32, device=cuda)
        hit = torch.full((B, (is_ + 7) // 8, (is_ + 31) // 32, 4), 9, dtype=torch.uint8, device=cuda)
        _lib.call("mr_render_flow_forward", P(d["v"]), P(d["fidx"]), P(d["cols"]), P(bg), 0, None, 0, 0.99999, P(rgb), P(alpha),
                  P(mask), None, P(wmap), P(fim), P(hit), P(work), wbytes, B, V, F0, 1, is_, 0.1, 100.0, 1e-3,
                  _lib.FLAG_SPARSE_TILES | flags, P(vid), bound, P(word) if bound else None, None, 0, 0, st)
        torch.cuda.synchronize()
        return [x.view(torch.int32) if x.dtype == torch.float32 else x for x in (rgb, alpha, mask, wmap, fim, vid, hit)]

    dense = run(0)
    assert int((dense[4] >= 0).sum()) > 100
    n_tiles = None
    for bound, flags in ((1, 0), (-1, 0), (10 ** 6, 0), (7, 512 << 8), (64, (512 | 1024) << 8)):
        got = run(bound, flags)
        for a, b_, name in zip(dense, got, ("rgb", "alpha", "mask", "weights", "face_index_map", "vertex ids", "coverage bytes")):
            assert torch.equal(a, b_), f"{name} differs with tile_bound={bound} flags={flags}"
        n = int(word[0])
        assert n_tiles in (None, n)
        n_tiles = n
    total = B * ((is_ + 7) // 8) * ((is_ + 31) // 32)
    assert 8 < n_tiles < total  # (a guess of 1 launches 8 workgroups: the rest of the list went through the overflow path)
    # every tile that reports coverage is on the list (the list also holds tiles whose candidates cover nothing)
    assert n_tiles >= int((dense[6].view(B, -1, 4).amax(2) > 0).sum())
    run(n_tiles)  # the guess a caller makes from the previous call
    assert int(word[0]) == n_tiles


def test_flow_forward_tile_list_of_an_empty_mesh(cuda):
    """num_faces = 0 with a listed launch: the per-face pass (which clears the list counters) does not run, so the entry
    point clears them itself -- on a workspace full of garbage the result is the background everywhere, an empty list and
    zero coverage bytes, for every tile_bound."""
    from handobjectconsist_amd import _lib

    B, is_, V = 2, 64, 5
    P, st = _lib.ptr, _lib.stream_ptr(cuda)
    f32 = dict(dtype=torch.float32, device=cuda)
    v, cols = torch.randn((B, V, 3), **f32), torch.randn((B, V, 3), **f32)
    fidx = torch.zeros((B, 1, 3), dtype=torch.int32, device=cuda)
    bg = torch.tensor([0.25, 0.5, 0.75], **f32)
    wbytes = int(_lib.load().mr_render_workspace_bytes(B, 2, is_))
    word = torch.zeros(1, dtype=torch.int32).pin_memory()
    for bound in (0, -1, 1, 10 ** 6):
        work = torch.full((wbytes,), 0xFF, dtype=torch.uint8, device=cuda)
        rgb = torch.full((B, 3, is_, is_), float("nan"), **f32)
        alpha, mask = torch.full((B, is_, is_), float("nan"), **f32), torch.full((B, is_, is_), float("nan"), **f32)
        wmap = torch.full((B, is_, is_, 3), float("nan"), **f32)
        fim = torch.full((B, is_, is_), -7, dtype=torch.int32, device=cuda)
        vid = torch.full((B, is_, is_, 3), -7, dtype=torch.int32, device=cuda)
        hit = torch.full((B, (is_ + 7) // 8, (is_ + 31) // 32, 4), 9, dtype=torch.uint8, device=cuda)
        word[0] = 123
        _lib.call("mr_render_flow_forward", P(v), P(fidx), P(cols), P(bg), 0, None, 0, 0.99999, P(rgb), P(alpha), P(mask), None,
                  P(wmap), P(fim), P(hit), P(work), wbytes, B, V, 0, 1, is_, 0.1, 100.0, 1e-3, _lib.FLAG_SPARSE_TILES, P(vid),
                  bound, P(word) if bound else None, None, 0, 0, st)
        torch.cuda.synchronize()
        assert int(hit.max()) == 0, f"coverage bytes, tile_bound={bound}"
        if bound:
            assert int(word[0]) == 0, f"list length {int(word[0])}, tile_bound={bound}"
        # nothing is rendered: a sparse launch writes no pixel (the coverage bytes say so); nothing may be garbage either
        assert int((fim != -7).sum()) == 0 or int((fim[fim != -7] != -1).sum()) == 0


def test_flow_forward_clears_the_backwards_output_buffer(cuda):
    """mr_render_flow_forward's zero_fill + MR_FLAG_OUTPUT_ZEROED of mr_render_flow_backward: the binning pass clears the
    (poisoned) gradient buffer, the backward adds into it without its own memset and gives what the self-clearing call
    gives; through autograd a second backward of the same node (retain_graph) falls back to the self-clearing call."""
    from handobjectconsist_amd import _lib
    from handobjectconsist_amd.neurender.renderer import Renderer
    from handobjectconsist_amd.warping import opticalflow

    B, is_ = 3, 96
    d = _vc_abi_case(cuda, B, is_, 44)
    P, st = _lib.ptr, _lib.stream_ptr(cuda)
    f32 = dict(dtype=torch.float32, device=cuda)
    V, F0 = d["V"], d["F0"]
    bg = torch.zeros(3, **f32)
    wbytes = int(_lib.load().mr_render_workspace_bytes(B, 2 * F0, is_))
    work = torch.empty((wbytes,), dtype=torch.uint8, device=cuda)
    rgb, alpha, mask = torch.empty((B, 3, is_, is_), **f32), torch.empty((B, is_, is_), **f32), torch.empty((B, is_, is_), **f32)
    wmap = torch.empty((B, is_, is_, 3), **f32)
    fim = torch.empty((B, is_, is_), dtype=torch.int32, device=cuda)
    vid = torch.empty((B, is_, is_, 3), dtype=torch.int32, device=cuda)
    hit = torch.empty((B, (is_ + 7) // 8, (is_ + 31) // 32, 4), dtype=torch.uint8, device=cuda)
    gbuf = torch.full((B, V, 3), float("nan"), **f32)
    _lib.call("mr_render_flow_forward", P(d["v"]), P(d["fidx"]), P(d["cols"]), P(bg), 0, None, 0, 0.99999, P(rgb), P(alpha), P(mask),
              None, P(wmap), P(fim), P(hit), P(work), wbytes, B, V, F0, 1, is_, 0.1, 100.0, 1e-3, _lib.FLAG_SPARSE_TILES, P(vid), -1,
              None, P(gbuf), int(gbuf.numel()), 0, st)
    assert float(gbuf.abs().max()) == 0.0 and not torch.isnan(gbuf).any()
    g = torch.Generator(device="cpu").manual_seed(9)
    g_rgb = torch.randn((B, 3, is_, is_), generator=g).to(cuda)
    ref = torch.full((B, V, 3), float("nan"), **f32)
    for out, flags in ((ref, 0), (gbuf, _lib.FLAG_OUTPUT_ZEROED)):
        _lib.call("mr_render_flow_backward", None, None, P(fim), P(hit), P(wmap), None, P(g_rgb), None, None, None, None, B, None,
                  is_, is_, P(out), B, V, F0, 1, is_, 1e-3, flags, P(vid), 0, None, st)
    assert float(ref.abs().sum()) > 0
    assert_close(gbuf.cpu().numpy(), ref.cpu().numpy(), 1e-5, 1e-6 * float(ref.abs().max()), "gradient in the pre-cleared buffer")

    # autograd: two backward passes through one stacked node
    s = synth.random_scene(2, seed=3, image_size=64)
    ren = Renderer(image_size=64, R=torch.eye(3, device=cuda)[None], t=torch.zeros(1, 3, device=cuda), K=torch.ones(1, 3, 3, device=cuda),
                   orig_size=64, anti_aliasing=False, fill_back=True, near=0.1, no_light=True)
    v1 = t(s["verts1"], cuda).requires_grad_(True)
    flows = opticalflow.get_opticalflow([v1, t(s["verts2"], cuda)], t(s["faces"], cuda), [t(s["K1"], cuda), t(s["K2"], cuda)], ren,
                                        orig_img_size=(64, 64), ignore_face_idxs=synth.HAND_IGNORE_FACES)
    gf = torch.randn_like(flows[0])
    (g1,) = torch.autograd.grad((flows[0] * gf).sum(), v1, retain_graph=True)
    (g2,) = torch.autograd.grad((flows[0] * gf).sum(), v1)
    assert float(g1.abs().sum()) > 0
    assert_close(g2.cpu().numpy(), g1.cpu().numpy(), 1e-5, 1e-6 * float(g1.abs().max()), "second backward of the node")


@pytest.mark.parametrize("table", [(0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0)])
def test_vertex_colour_kernels_follow_the_texel_table(cuda, table):
    """Which texel of libyana's 2x2x2 vertex-colour texture holds which vertex is an ASSUMPTION (source absent): it is
    a table (utils/textutils.TEXEL_VERTEX -> the C-ABI's texel_layout), not code.  For every permutation the fused
    vertex-colour kernels (full-output render, flow-mode render with and without per-pixel records, their backward
    passes) agree with the generic path on the materialised textures of the same table -- with fill-back, whose
    reversed copies mirror the texture axes."""
    from handobjectconsist_amd.neurender.renderer import Renderer
    from handobjectconsist_amd.utils import textutils
    from handobjectconsist_amd.warping import opticalflow

    B, is_ = 2, 64
    s = synth.random_scene(B, seed=12, image_size=is_)
    ren = Renderer(image_size=is_, R=torch.eye(3, device=cuda)[None], t=torch.zeros(1, 3, device=cuda), K=torch.ones(1, 3, 3, device=cuda),
                   orig_size=is_, anti_aliasing=False, fill_back=True, near=0.1, no_light=True)
    faces, K1, K2 = t(s["faces"], cuda), t(s["K1"], cuda), t(s["K2"], cuda)
    g = torch.Generator().manual_seed(2)
    cols0 = torch.randn(B, s["verts1"].shape[1], 3, generator=g).to(cuda)
    g_rgb = torch.randn(B, 3, is_, is_, generator=g).to(cuda)
    saved = (textutils.TEXEL_VERTEX, opticalflow.USE_VERTEX_COLOR_RENDER, opticalflow.USE_PIXEL_RECORDS)
    textutils.TEXEL_VERTEX = table
    try:
        # full-output render: fused vertex-colour kernels vs materialised textures
        outs = []
        for fused in (True, False):
            cols = cols0.clone().requires_grad_(True)
            if fused:
                out = ren.render_vertex_colors(t(s["verts1"], cuda), faces, cols, K=K1)
            else:
                out = ren(t(s["verts1"], cuda), faces, textutils.batch_vertex_textures(faces, cols), K=K1, detach_renders=True)
            (out["rgb"] * g_rgb).sum().backward()
            outs.append((out["rgb"].detach(), out["face_index_map"], cols.grad))
        assert torch.equal(outs[0][1], outs[1][1])
        assert float((outs[0][0] - outs[1][0]).abs().max()) <= 1e-6 * float(outs[1][0].abs().max())
        assert_close(outs[0][2].cpu().numpy(), outs[1][2].cpu().numpy(), 1e-4, 1e-5 * float(outs[1][2].abs().max()), "d rgb / d colours")
        if table != (0, 1, 2):  # ... and the table matters: the identity layout gives another image
            textutils.TEXEL_VERTEX = (0, 1, 2)
            other = ren.render_vertex_colors(t(s["verts1"], cuda), faces, cols0, K=K1)["rgb"]
            textutils.TEXEL_VERTEX = table
            assert float((other - outs[0][0]).abs().max()) > 1e-3
        # get_opticalflow: stacked training node (records / stored maps) vs the op-by-op path on materialised textures
        res = []
        for vc, rec in ((True, True), (True, False), (False, True)):
            opticalflow.USE_VERTEX_COLOR_RENDER, opticalflow.USE_PIXEL_RECORDS = vc, rec
            v1, v2 = t(s["verts1"], cuda).requires_grad_(True), t(s["verts2"], cuda).requires_grad_(True)
            flows = opticalflow.get_opticalflow([v1, v2], faces, [K1, K2], ren, orig_img_size=(is_, is_),
                                                ignore_face_idxs=synth.HAND_IGNORE_FACES)
            gf = torch.Generator().manual_seed(4)
            w12, w21 = torch.randn(flows[0].shape, generator=gf).to(cuda), torch.randn(flows[1].shape, generator=gf).to(cuda)
            ((flows[0] * w12).sum() + (flows[1] * w21).sum()).backward()
            res.append((flows[0].detach(), flows[1].detach(), v1.grad, v2.grad))
        for got in res[:2]:
            for a, b_, name in zip(got, res[2], ("flow12", "flow21", "d / d verts1", "d / d verts2")):
                if name.startswith("flow"):
                    assert int(((a != 0) != (b_ != 0)).sum()) <= 4, name  # (independent projections: README caveat)
                    both = (a != 0) & (b_ != 0)
                    assert float(((a - b_) * both).abs().max()) < 5e-3 and float((a - b_)[both].abs().median()) < 1e-5, name
                else:
                    assert float((a - b_).norm() / b_.norm()) < 5e-2, name
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])  # records vs stored maps: same flows
        assert_close(res[0][2].cpu().numpy(), res[1][2].cpu().numpy(), 1e-5, 1e-6 * float(res[1][2].abs().max()), "records vs maps")
    finally:
        textutils.TEXEL_VERTEX, opticalflow.USE_VERTEX_COLOR_RENDER, opticalflow.USE_PIXEL_RECORDS = saved


def test_shared_reciprocal_division_is_the_ieee_quotient(cuda):
    """rcp_refined + div_refined (the 3 + 5 n instruction sequence behind the tile kernel's divisions by a common
    denominator) against the compiler's `/` on the GPU and numpy's on the CPU: the same bits for 6 M operand pairs over
    the admitted ranges -- magnitudes 2^-30 .. 2^30, numerators also 0 and down to 2^-62, mantissas next to powers of two."""
    from handobjectconsist_amd import _lib

    rng = np.random.default_rng(7)
    n = 6_000_000
    sign = lambda: rng.choice(np.array([-1.0, 1.0], np.float32), n)
    b = (np.exp2(rng.uniform(-30, 30, n)).astype(np.float32) * sign())
    a = (np.exp2(rng.uniform(-62, 30, n)).astype(np.float32) * sign())
    third = n // 3
    a[:third] = (np.exp2(rng.uniform(-3, 3, third)).astype(np.float32) * sign()[:third])  # quotients of comparable numbers
    edge = rng.integers(0, 6, n)
    pow2 = lambda x: np.exp2(np.round(np.log2(np.abs(x)))).astype(np.float32)
    b = np.where(edge == 0, np.nextafter(pow2(b), np.float32(0)), b).astype(np.float32)          # mantissa all ones
    b = np.where(edge == 1, np.nextafter(pow2(b), np.float32(np.inf)), b).astype(np.float32)     # 1 + ulp
    a = np.where(edge == 2, np.nextafter(pow2(a), np.float32(0)), a).astype(np.float32)
    a[::97] = 0.0
    a[1::97] = -0.0
    ta, tb = t(a, cuda), t(b, cuda)
    refined, plain = torch.empty_like(ta), torch.empty_like(ta)
    _lib.call("mr_selftest_division", _lib.ptr(ta), _lib.ptr(tb), _lib.ptr(refined), _lib.ptr(plain), n, _lib.stream_ptr(cuda))
    want = (a / b).astype(np.float32)
    assert np.array_equal(plain.cpu().numpy().view(np.uint32), want.view(np.uint32)), "the GPU's `/` is the IEEE quotient"
    bad = refined.cpu().numpy().view(np.uint32) != want.view(np.uint32)
    assert not bad.any(), (int(bad.sum()), a[bad][:4], b[bad][:4])


@pytest.mark.parametrize("case", ["bench scene", "small faces", "mixed magnitudes"])
def test_tile_kernel_divisions_shared_vs_plain(cuda, case):
    """The forward tile kernel with its shared-reciprocal division paths (faces the per-face pass admits) and with the
    plain `/` everywhere (profiling switch 4096): every output plane, the face index map, the per-pixel records and the
    coverage bytes are the same bits."""
    from handobjectconsist_amd import _lib
    from handobjectconsist_amd.neurender import nr_ops

    f32 = dict(dtype=torch.float32, device=cuda)
    if case == "bench scene":
        B, is_ = 6, 256
        s = synth.random_scene(B, seed=0, image_size=is_)
        v = nr_ops.projection(t(s["verts1"], cuda), t(s["K1"], cuda), torch.eye(3, device=cuda)[None], torch.zeros(1, 3, device=cuda),
                              torch.zeros(1, 5, device=cuda), is_).contiguous()
        fidx = t(s["faces"], cuda).to(torch.int32).contiguous()
    else:
        B, is_ = 4, 96
        d = _vc_abi_case(cuda, B, is_, 51)
        v, fidx = d["v"].clone(), d["fidx"]
        if case == "mixed magnitudes":  # some depths / coordinates outside the admitted ranges: those faces keep `/`
            v[:, ::5, 2] *= 1e12
            v[:, 1::7, :2] *= 3e4
            v[:, 2::11, 2] = 1e-20
    V, F0 = v.shape[1], fidx.shape[1]
    cols = torch.randn(B, V, 3, generator=torch.Generator().manual_seed(1)).to(cuda)
    P, st = _lib.ptr, _lib.stream_ptr(cuda)
    bg = torch.zeros(3, **f32)
    wbytes = int(_lib.load().mr_render_workspace_bytes(B, 2 * F0, is_))
    work = torch.empty((wbytes,), dtype=torch.uint8, device=cuda)
    outs = []
    for dbg in (0, 4096):
        for records in (True, False):
            rgb = torch.zeros((B, 3, is_, is_), **f32)
            alpha, mask, depth = torch.zeros((B, is_, is_), **f32), torch.zeros((B, is_, is_), **f32), torch.zeros((B, is_, is_), **f32)
            wmap = torch.zeros((B, is_, is_, 3), **f32)
            fim = torch.full((B, is_, is_), -7, dtype=torch.int32, device=cuda)
            vid = torch.full((B, is_, is_, 3), -7, dtype=torch.int32, device=cuda)
            hit = torch.zeros((B, (is_ + 7) // 8, (is_ + 31) // 32, 4), dtype=torch.uint8, device=cuda)
            _lib.call("mr_render_flow_forward", P(v), P(fidx), P(cols), P(bg), 0, None, 0, 0.99999, P(rgb), P(alpha), P(mask),
                      None if records else P(depth), P(wmap), P(fim), P(hit), P(work), wbytes, B, V, F0, 1, is_, 0.1, 100.0, 1e-3,
                      _lib.FLAG_SPARSE_TILES | (dbg << 8), P(vid) if records else None, -1, None, None, 0, 0, st)
            outs.append([x.view(torch.int32) if x.dtype == torch.float32 else x for x in (rgb, alpha, mask, depth, wmap, fim, vid, hit)])
    assert int((outs[0][5] >= 0).sum()) > 200
    for fast, plain in ((outs[0], outs[2]), (outs[1], outs[3])):
        for a, b_, name in zip(fast, plain, ("rgb", "alpha", "mask", "depth", "weights", "face_index_map", "vertex ids", "coverage")):
            assert torch.equal(a, b_), f"{case}: {name} differs between the shared-reciprocal and the plain divisions"


def test_training_mode_textures_only(cuda):
    """detach_renders=True (warpbranch.py:65-66): only grad_textures is live.  The gather sums a
    face's pixels as four row-interleaved partial sums (fixed order), so it matches the serial
    oracle to fp32 round-off and is bit-reproducible run to run."""
    from handobjectconsist_amd.neurender import rasterize

    faces, tex = projected_faces(2, 128, 7)
    ref = R.rasterize_rgbad(faces, tex, 128, False, 0.1, 100, 1e-3, (0, 0, 0), num_threads=8, keep_saved=True)
    raster_g, img_g = _img_grads(ref["_saved"], 7)
    _, gt_ref = R.rasterize_backward(ref["_saved"], raster_g[0], None, None, num_threads=8)
    x_t = t(tex, cuda).requires_grad_(True)
    out = rasterize.rasterize_rgbad(t(faces, cuda), x_t, 128, False, 0.1, 100, 1e-3, (0, 0, 0))
    out["rgb"].backward(t(img_g[0], cuda))
    got = x_t.grad.cpu().numpy()
    assert np.abs(gt_ref).max() > 0
    assert_close(got, gt_ref, 1e-5, 1e-6 * np.abs(gt_ref).max(), "grad_textures (training mode)")
    x2 = t(tex, cuda).requires_grad_(True)
    out2 = rasterize.rasterize_rgbad(t(faces, cuda), x2, 128, False, 0.1, 100, 1e-3, (0, 0, 0))
    out2["rgb"].backward(t(img_g[0], cuda))
    assert torch.equal(x2.grad, x_t.grad), "gather backward is not deterministic"


def test_edge_cases(cuda):
    from handobjectconsist_amd.neurender import rasterize

    # coplanar duplicates (tie -> lowest index), near/far rejection, degenerate + NaN faces, odd size
    is_ = 37
    tri = np.array([[-0.8, -0.7, 1.0], [0.9, -0.6, 1.0], [0.1, 0.85, 1.0]], np.float32)
    # (the point face -- three vertices on one pixel-centre-lattice point -- passes every edge test of
    # upstream's inside test at every pixel and has a NaN depth: it must never win; found by tests/fuzz_parity.py)
    point = np.array([[-0.5, -0.5, 1.75], [-0.5, -0.5, 1.625], [-0.5, -0.5, 1.75]], np.float32)
    # collinear vertices far off screen: the pixel-space determinant rounds to a non-zero value, yet upstream's
    # edge tests accept the pixels exactly on the line, also BEYOND the vertices (bbox filter in face_box)
    line = np.array([[-18.75, 31.25, 2.75], [0.0, 0.0, 0.5], [-37.5, 62.5, 1.625]], np.float32)
    faces = np.stack([tri, tri, tri * [1, 1, 0.05], tri * [1, 1, 500.0], tri[[0, 0, 1]], tri * np.nan,
                      tri[::-1], point, point[::-1], line, line[::-1]])[None]
    tex = np.random.default_rng(0).uniform(0, 1, (1, faces.shape[1], 2, 2, 2, 3)).astype(np.float32)
    ref = R.rasterize_rgbad(faces, tex, is_, False, 0.1, 100, 1e-3, (0, 0, 0))
    out = rasterize.rasterize_rgbad(t(faces, cuda), t(tex, cuda), is_, False, 0.1, 100, 1e-3, (0, 0, 0))
    fim = out["face_index_map"].cpu().numpy()
    assert (fim != ref["face_index_map"]).sum() == 0
    assert set(np.unique(fim)) == {-1, 0}
    # the line face at the resolution where its line passes through pixel centres (also beyond its vertices)
    lf = np.stack([line, line[::-1]])[None]
    ltex = tex[:, :2]
    ref = R.rasterize_rgbad(lf, ltex, 274, False, 0.1, 100, 1e-3, (0, 0, 0))
    out = rasterize.rasterize_rgbad(t(lf, cuda), t(ltex, cuda), 274, False, 0.1, 100, 1e-3, (0, 0, 0))
    assert (ref["face_index_map"] >= 0).sum() >= 20 and (ref["face_index_map"][0, :130] >= 0).any()
    assert (out["face_index_map"].cpu().numpy() != ref["face_index_map"]).sum() == 0
    assert_close(out["rgb"].cpu().numpy(), ref["rgb"], 1e-6, 1e-6, "rgb")
    # empty batch / zero faces
    e = rasterize.rasterize_rgbad(torch.zeros((1, 0, 3, 3), device=cuda), torch.zeros((1, 0, 2, 2, 2, 3), device=cuda),
                                  16, False)
    assert (e["face_index_map"] == -1).all() and (e["alpha"] == 0).all() and (e["depth"] == 100).all()
    # anti-aliasing: 2x raster + average pool, maps stay at 2x
    faces2, tex2 = projected_faces(1, 64, 5)
    ref = R.rasterize_rgbad(faces2, tex2, 32, True, 0.1, 100, 1e-3, (0, 0, 0))
    out = rasterize.rasterize_rgbad(t(faces2, cuda), t(tex2, cuda), 32, True, 0.1, 100, 1e-3, (0, 0, 0))
    assert out["face_index_map"].shape == (1, 64, 64) and out["rgb"].shape == (1, 3, 32, 32)
    assert_close(out["rgb"].cpu().numpy(), ref["rgb"], 1e-6, 1e-6, "aa rgb")
    assert_close(out["alpha"].cpu().numpy(), ref["alpha"], 0, 1e-7, "aa alpha")
    # CPU tensors are rejected like the reference (rasterize.py:346-347)
    with pytest.raises(TypeError):
        rasterize.rasterize_rgbad(torch.zeros((1, 1, 3, 3)), torch.zeros((1, 1, 2, 2, 2, 3)), 8, False)


def test_silhouette_config1(cuda):
    """BASELINE config 1: single 1538-face hand-like mesh, 64x64 AA silhouette + depth modes."""
    from handobjectconsist_amd.neurender.renderer import Renderer

    s = synth.random_scene(1, seed=11, image_size=64)
    verts, fidx = s["hand_verts1"], s["hand_faces"][None, :1538]
    ren = Renderer(image_size=64, anti_aliasing=True, fill_back=True, camera_mode="projection",
                   K=t(s["K1"], cuda), R=torch.eye(3, device=cuda)[None], t=torch.zeros(1, 3, device=cuda),
                   orig_size=64)
    sil = ren(t(verts, cuda), t(fidx, cuda), mode="silhouettes").cpu().numpy()
    dep = ren(t(verts, cuda), t(fidx, cuda), mode="depth").cpu().numpy()
    f2, _ = R.fill_back(fidx)
    v = R.nr_projection(verts, s["K1"], REN_KW["R"], REN_KW["t"], REN_KW["dist_coeffs"], 64)
    ref = R.rasterize_rgbad(R.nr_vertices_to_faces(v, f2), None, 64, True, 0.1, 100, 1e-4, None, False, True, True)
    assert sil.shape == (1, 64, 64) and sil.sum() > 20
    assert_close(sil, ref["alpha"], 0, 1e-7, "silhouette")
    assert_close(dep, ref["depth"], 1e-6, 0, "depth")


@pytest.mark.parametrize("fill_back,aa", [(True, False), (False, False), (True, True)])
def test_vertex_color_render_matches_generic_path_and_oracle(cuda, fill_back, aa):
    """Renderer.render_vertex_colors == Renderer.render(batch_vertex_textures(...), detach_renders=True):
    identical images / maps, same gradient w.r.t. the vertex colours; and == the oracle chain."""
    from handobjectconsist_amd.neurender.renderer import Renderer
    from handobjectconsist_amd.utils import textutils

    B, is_ = 3, 96
    s = synth.random_scene(B, seed=31, image_size=is_)
    rng = np.random.default_rng(5)
    cols_np = rng.uniform(-3, 3, (B, s["verts1"].shape[1], 3)).astype(np.float32)
    ren = Renderer(image_size=is_, R=torch.eye(3, device=cuda)[None], t=torch.zeros(1, 3, device=cuda),
                   K=torch.ones(1, 3, 3, device=cuda), orig_size=is_, anti_aliasing=aa, fill_back=fill_back, near=0.1,
                   no_light=True)
    verts, fidx, K = t(s["verts1"], cuda), t(s["faces"], cuda), t(s["K1"], cuda)
    c1 = t(cols_np, cuda).requires_grad_(True)
    out_vc = ren.render_vertex_colors(verts, fidx, c1, K=K)
    c2 = t(cols_np, cuda).requires_grad_(True)
    out_gen = ren(verts, fidx, textutils.batch_vertex_textures(fidx, c2), K=K, detach_renders=True)
    for k in ("rgb", "alpha", "depth", "weight_map", "face_inv_map"):
        assert torch.equal(out_vc[k], out_gen[k]), k
    assert torch.equal(out_vc["face_index_map"], out_gen["face_index_map"])
    g = torch.randn_like(out_vc["rgb"])
    out_vc["rgb"].backward(g)
    out_gen["rgb"].backward(g)
    assert_close(c1.grad.cpu().numpy(), c2.grad.cpu().numpy(), 1e-4, 1e-5 * float(c2.grad.abs().max()), "grad colours")
    assert c2.grad.abs().max() > 0
    # oracle chain
    tex = R.batch_vertex_textures(s["faces"], cols_np)
    ref = R.render(s["verts1"], s["faces"], tex, s["K1"], REN_KW["R"], REN_KW["t"], REN_KW["dist_coeffs"], is_, is_,
                   anti_aliasing=aa, fill_back_=fill_back, near=0.1, far=100, eps=1e-3, num_threads=8)
    assert (out_vc["face_index_map"].cpu().numpy() != ref["face_index_map"]).sum() == 0
    # (vertices are projected by torch on the GPU here and by numpy in the oracle chain: the face
    # coordinates differ in the last bit, hence 1e-4 of the colour range instead of 1e-6)
    assert_close(out_vc["rgb"].detach().cpu().numpy(), ref["rgb"], 1e-4, 3e-4, "rgb vs oracle")


@pytest.mark.parametrize("ts", [3, 4])
def test_generic_texture_size_and_per_sample_background(cuda, ts):
    """Textures larger than 2x2x2 (per-pixel-atomics backward on recomputed sampling weights) and a
    [B,3] background colour (rasterize.py:254-258)."""
    from handobjectconsist_amd.neurender import rasterize

    B, is_ = 2, 64
    faces, _ = projected_faces(B, is_, 17)
    rng = np.random.default_rng(ts)
    tex = rng.uniform(-1, 1, (B, faces.shape[1], ts, ts, ts, 3)).astype(np.float32)
    bg = [[0.1, 0.2, 0.3], [0.9, 0.8, 0.7]]
    ref = R.rasterize_rgbad(faces, tex, is_, False, 0.1, 100, 1e-3, bg, num_threads=8, keep_saved=True)
    x_t = t(tex, cuda).requires_grad_(True)
    out = rasterize.rasterize_rgbad(t(faces, cuda), x_t, is_, False, 0.1, 100, 1e-3, bg)
    assert (out["face_index_map"].cpu().numpy() != ref["face_index_map"]).sum() == 0
    assert_close(out["rgb"].detach().cpu().numpy(), ref["rgb"], 1e-6, 1e-6, "rgb")
    assert np.allclose(out["rgb"].detach().cpu().numpy()[1, :, 0, 0], bg[1])
    raster_g, img_g = _img_grads(ref["_saved"], ts)
    _, gt_ref = R.rasterize_backward(ref["_saved"], raster_g[0], None, None, num_threads=8)
    out["rgb"].backward(t(img_g[0], cuda))
    assert_close(x_t.grad.cpu().numpy(), gt_ref, 1e-4, 1e-5 * np.abs(gt_ref).max(), "grad_textures")


def _vc_abi_case(cuda, B, is_, seed, n_extra_verts=0):
    """Forward of the vertex-colour path through the C-ABI; returns device buffers + oracle inputs."""
    from handobjectconsist_amd import _lib

    s = synth.random_scene(B, seed=seed, image_size=is_)
    v = R.nr_projection(s["verts1"], s["K1"], REN_KW["R"], REN_KW["t"], REN_KW["dist_coeffs"], is_)
    if n_extra_verts:  # unreferenced vertices: only the size of the colour table changes
        v = np.concatenate([v, np.ones((B, n_extra_verts, 3), np.float32)], 1)
    fidx = s["faces"].astype(np.int32)
    rng = np.random.default_rng(seed)
    cols = rng.uniform(-2, 2, (B, v.shape[1], 3)).astype(np.float32)
    V, F0 = v.shape[1], fidx.shape[1]
    P, st = _lib.ptr, _lib.stream_ptr(cuda)
    f32 = dict(dtype=torch.float32, device=cuda)
    d = dict(v=t(v, cuda), fidx=t(fidx, cuda), cols=t(cols, cuda), rgb=torch.empty((B, 3, is_, is_), **f32),
             alpha=torch.empty((B, is_, is_), **f32), depth=torch.empty((B, is_, is_), **f32),
             fim=torch.empty((B, is_, is_), dtype=torch.int32, device=cuda), wmap=torch.empty((B, is_, is_, 3), **f32))
    wbytes = int(_lib.load().mr_render_workspace_bytes(B, 2 * F0, is_))
    work = torch.empty((wbytes,), dtype=torch.uint8, device=cuda)
    bg = torch.zeros(3, **f32)
    _lib.call("mr_render_vc_forward", P(d["v"]), P(d["fidx"]), P(d["cols"]), P(bg), 0, P(d["rgb"]), P(d["alpha"]),
              P(d["depth"]), P(d["fim"]), P(d["wmap"]), P(work), wbytes, B, V, F0, 1, is_, 0.1, 100.0, 1e-3, 1, 1, 1, 0, 0, st)
    d.update(B=B, V=V, F0=F0, is_=is_, v_np=v, fidx_np=fidx, cols_np=cols)
    return d


def _vc_backward(d, g_rgb_img, mode):
    """mode: 'stored' (forward maps), 'recompute' (NULL maps), 'gather' (face-parallel kernel), 'tiles' (the
    tile-persistent kernel of mr_render_flow_backward, every tile visited), 'tiles_hit' (... with coverage bytes)."""
    from handobjectconsist_amd import _lib

    P = _lib.ptr
    out = torch.full((d["B"], d["V"], 3), float("nan"), dtype=torch.float32, device=g_rgb_img.device)
    if mode in ("tiles", "tiles_hit"):
        hit = None
        if mode == "tiles_hit":  # what mr_render_flow_forward writes: per (32x8 tile, row pair) "anything covered"
            B, is_ = d["B"], d["is_"]
            ty, tx = (is_ + 7) // 8, (is_ + 31) // 32
            cov = torch.zeros((B, ty * 8, tx * 32), dtype=torch.bool, device=g_rgb_img.device)
            cov[:, :is_, :is_] = d["fim"] >= 0
            hit = cov.reshape(B, ty, 4, 2, tx, 32).permute(0, 1, 4, 2, 3, 5).reshape(B, ty, tx, 4, 64).any(-1).to(torch.uint8).contiguous()
        _lib.call("mr_render_flow_backward", P(d["v"]), P(d["fidx"]), P(d["fim"]), P(hit), P(d["wmap"]), P(d["depth"]),
                  P(g_rgb_img), None, None, None, None, 0, None, 0, 0, P(out), d["B"], d["V"], d["F0"], 1, d["is_"], 1e-3, 0,
                  None, 0, None, _lib.stream_ptr(g_rgb_img.device))
        return out.cpu().numpy()
    stored = mode == "stored"
    _lib.call("mr_render_vc_backward", P(d["v"]), P(d["fidx"]), P(d["fim"]), P(d["wmap"]) if stored else None,
              P(d["depth"]) if stored else None, P(g_rgb_img), P(out), d["B"], d["V"], d["F0"], 1, d["is_"], 1e-3,
              (32 << 8) if mode == "gather" else 0, 0, _lib.stream_ptr(g_rgb_img.device))
    return out.cpu().numpy()


def _vc_backward_oracle(d, g_rgb_raster):
    """oracle kernel E on the materialised textures, then the adjoints of the fill-back concatenation
    and of batch_vertex_textures in numpy (float64 accumulation)."""
    B, V, F0, is_ = d["B"], d["V"], d["F0"], d["is_"]
    tex = R.batch_vertex_textures(d["fidx_np"], d["cols_np"])
    f2, tex2 = R.fill_back(d["fidx_np"], tex)
    saved = R.rasterize_forward(R.nr_vertices_to_faces(d["v_np"], f2), tex2, is_, 0.1, 100, 1e-3, (0, 0, 0),
                                num_threads=8)
    _, gt = R.rasterize_backward(saved, g_rgb_raster, None, None, num_threads=8)
    gt = gt.astype(np.float64)
    g_tex = gt[:, :F0] + gt[:, F0:].transpose(0, 1, 4, 3, 2, 5)  # texel (i,j,k) of the copy is (k,j,i)
    g_cols = np.zeros((B, V, 3))
    for b in range(B):
        np.add.at(g_cols[b], d["fidx_np"][b, :, 0], g_tex[b, :, 1, 0, 0])
        np.add.at(g_cols[b], d["fidx_np"][b, :, 1], g_tex[b, :, 0, 1, 0])
        np.add.at(g_cols[b], d["fidx_np"][b, :, 2], g_tex[b, :, 0, 0, 1])
    return g_cols, saved


@pytest.mark.parametrize("B,is_,seed", [(2, 64, 0), (3, 100, 1), (2, 256, 2)])
def test_vc_backward_kernels_match_oracle(cuda, B, is_, seed):
    """All three implementations of the vertex-colour backward (pixel-parallel scatter on the stored
    maps / on recomputed barycentrics, face-parallel gather) against oracle kernel E."""
    d = _vc_abi_case(cuda, B, is_, seed)
    rng = np.random.default_rng(seed + 7)
    g_raster = rng.standard_normal((B, is_, is_, 3)).astype(np.float32)
    # region-dependent magnitudes: the fixed-point scale of the scatter kernel is per region
    g_raster *= np.exp2(rng.integers(-30, 30, (B, is_ // 16 + 1, 1, 1))).astype(np.float32).repeat(16, 1)[:, :is_]
    g_img = t(g_raster.transpose(0, 3, 1, 2)[:, :, ::-1], cuda)
    ref, saved = _vc_backward_oracle(d, g_raster)
    assert (d["fim"].cpu().numpy() == saved["face_index_map"]).all()
    assert np.abs(ref).max() > 0
    # per-vertex tolerance: 1e-4 of the vertex's own absolute contributions would need the oracle's
    # partial sums; use 1e-4 relative + 1e-6 of the largest gradient the vertex's image region saw
    atol = 1e-6 * np.abs(ref).max()
    for mode in ("stored", "recompute", "gather", "tiles", "tiles_hit"):
        got = _vc_backward(d, g_img, mode)
        assert np.isfinite(got).all(), mode
        assert_close(got, ref, 1e-4, atol, f"grad_vcolors[{mode}]")
    a, b_ = _vc_backward(d, g_img, "stored"), _vc_backward(d, g_img, "recompute")
    assert_close(a, b_, 1e-5, 1e-7 * np.abs(ref).max(), "stored vs recompute")


def test_vc_backward_edge_cases(cuda):
    d = _vc_abi_case(cuda, 2, 96, 5)
    B, is_ = d["B"], d["is_"]
    zero = torch.zeros((B, 3, is_, is_), dtype=torch.float32, device=cuda)
    for mode in ("stored", "recompute", "gather", "tiles", "tiles_hit"):
        assert (_vc_backward(d, zero, mode) == 0).all(), mode  # also: the output is zeroed by the call
    # a non-finite gradient on a covered pixel reaches exactly the three vertices of the winning face
    fim = d["fim"].cpu().numpy()
    ys, xs = np.nonzero(fim[0] >= 0)
    y, x = int(ys[len(ys) // 2]), int(xs[len(xs) // 2])
    fn = int(fim[0, y, x])
    tri = d["fidx_np"][0, fn % d["F0"]]
    g = torch.randn((B, 3, is_, is_), dtype=torch.float32, device=cuda)
    g[0, 1, is_ - 1 - y, x] = float("inf")
    for mode in ("stored", "recompute", "gather", "tiles", "tiles_hit"):
        got = _vc_backward(d, g, mode)
        bad = np.argwhere(~np.isfinite(got))
        assert len(bad) > 0 and set(bad[:, 0]) == {0} and set(bad[:, 1]) <= set(tri.tolist()) and set(bad[:, 2]) == {1}, mode
    # denormal-sized and huge gradients keep their relative accuracy (per-region scale)
    for scale in (1e-38, 1e30):
        gs = torch.randn((B, 3, is_, is_), dtype=torch.float32, device=cuda) * scale
        a, b_ = _vc_backward(d, gs, "stored"), _vc_backward(d, gs, "gather")
        assert_close(a, b_, 1e-4, 1e-6 * np.abs(b_).max(), f"scale {scale}")
        assert_close(_vc_backward(d, gs, "tiles_hit"), b_, 1e-4, 1e-6 * np.abs(b_).max(), f"tiles, scale {scale}")
        assert np.abs(b_).max() > 0
    # a colour table too large for LDS falls back to the gather kernel
    big = _vc_abi_case(cuda, 1, 64, 6, n_extra_verts=4000)
    gb = torch.randn((1, 3, 64, 64), dtype=torch.float32, device=cuda)
    a, b_ = _vc_backward(big, gb, "stored"), _vc_backward(big, gb, "gather")
    assert_close(a, b_, 1e-6, 1e-7 * np.abs(b_).max(), "large V")
    assert (a[:, -4000:] == 0).all()


@pytest.mark.parametrize("B,is_,H,W", [(2, 96, 96, 96), (3, 128, 72, 128), (2, 40, 27, 40)])
def test_flow_backward_flow_space_gradient(cuda, B, is_, H, W):
    """mr_render_flow_backward fed with the FLOW-space gradient + epilogue masks == mr_flow_finalize_backward
    followed by mr_render_vc_backward (two halves with different mask_x, as for a stacked frame pair), incl. a
    non-square crop and an image size that is not a multiple of the tile."""
    from handobjectconsist_amd import _lib

    d = _vc_abi_case(cuda, 2 * B, is_, 11)
    B2 = 2 * B
    P, st = _lib.ptr, _lib.stream_ptr(cuda)
    g = torch.Generator(device="cpu").manual_seed(3)
    gf = torch.randn((B2, H, W, 2), generator=g).to(cuda)
    m_pre = (torch.rand((B2, is_, is_), generator=g) < 0.8).float().to(cuda)
    m_lo = (torch.rand((B, is_, is_), generator=g) < 0.8).float().to(cuda)
    m_hi = torch.rand((B, is_, is_), generator=g).to(cuda)  # raw alpha may be any float (Q4)
    occl = (torch.rand((B2, is_, is_), generator=g) < 0.9).float().to(cuda)
    grad_rgb = torch.empty((B2, 3, is_, is_), dtype=torch.float32, device=cuda)
    for lo, mx in ((0, m_lo), (B, m_hi)):
        _lib.call("mr_flow_finalize_backward", P(gf[lo:lo + B]), P(m_pre[lo:lo + B]), P(mx), P(occl[lo:lo + B]),
                  P(grad_rgb[lo:lo + B]), B, is_, H, W, st)
    ref = _vc_backward(d, grad_rgb, "stored")
    out = torch.full((B2, d["V"], 3), float("nan"), dtype=torch.float32, device=cuda)
    _lib.call("mr_render_flow_backward", P(d["v"]), P(d["fidx"]), P(d["fim"]), None, P(d["wmap"]), P(d["depth"]), None,
              P(gf), P(m_pre), P(m_lo), P(m_hi), B, P(occl), H, W, P(out), B2, d["V"], d["F0"], 1, is_, 1e-3, 0, None, 0, None, st)
    got = out.cpu().numpy()
    assert np.abs(ref).max() > 0 and (got[:, :, 2] == 0).all()
    assert_close(got, ref, 1e-5, 1e-6 * np.abs(ref).max(), "flow-space gradient form")


def test_vertex_colour_render_adjoint_at_metric_size(cuda):
    """BASELINE metric config (B=64, 256x256, 7104 faces): no oracle run at this size, but the render
    is LINEAR in the vertex colours for fixed geometry (background 0), so its backward must be the exact
    transpose: <render(c), g> == <c, backward(g)>; plus linearity and coverage sanity."""
    d = _vc_abi_case(cuda, 64, 256, 0)
    from handobjectconsist_amd import _lib

    B, V, F0, is_ = d["B"], d["V"], d["F0"], d["is_"]
    P, st = _lib.ptr, _lib.stream_ptr(cuda)
    f32 = dict(dtype=torch.float32, device=cuda)
    wbytes = int(_lib.load().mr_render_workspace_bytes(B, 2 * F0, is_))
    work = torch.empty((wbytes,), dtype=torch.uint8, device=cuda)
    bg = torch.zeros(3, **f32)

    def render(cols):
        rgb = torch.empty((B, 3, is_, is_), **f32)
        _lib.call("mr_render_vc_forward", P(d["v"]), P(d["fidx"]), P(cols), P(bg), 0, P(rgb), P(d["alpha"]), P(d["depth"]),
                  P(d["fim"]), P(d["wmap"]), P(work), wbytes, B, V, F0, 1, is_, 0.1, 100.0, 1e-3, 1, 1, 1, 0, 0, st)
        return rgb

    g = torch.Generator(device="cpu").manual_seed(0)
    c1 = torch.randn((B, V, 3), generator=g).to(cuda)
    c2 = torch.randn((B, V, 3), generator=g).to(cuda)
    gr = torch.randn((B, 3, is_, is_), generator=g).to(cuda)
    r1, r2, r12 = render(c1), render(c2), render(c1 + 2 * c2)
    assert float((r12 - (r1 + 2 * r2)).abs().max()) <= 1e-4 * float(r12.abs().max())  # linearity
    cov = (d["fim"] >= 0)
    assert 0.05 < float(cov.float().mean()) < 0.5
    assert float(r1[(~cov).unsqueeze(1).expand_as(r1).flip(2)].abs().max()) == 0.0  # background stays 0 (image is flipped)
    back = torch.from_numpy(_vc_backward(d, gr, "stored")).to(cuda)
    lhs = float((r1.double() * gr.double()).sum())
    rhs = float((c1.double() * back.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), abs(rhs), float(r1.abs().sum()) * 1e-3), (lhs, rhs)
    # gradients only reach vertices of faces that own a pixel
    owners = torch.zeros((B, V), dtype=torch.bool, device=cuda)
    fim = d["fim"].long()
    for b in range(0, B, 16):
        fn = torch.unique(fim[b][fim[b] >= 0]) % F0
        owners[b, d["fidx"][b, fn].long().flatten()] = True
        assert (back[b][~owners[b]] == 0).all()


def test_integration_md_stub_runs_the_reference_call_sequence(cuda):
    """The `neural_renderer.cuda.rasterize` replacement printed in INTEGRATION.md (section B), executed
    verbatim and driven with the reference's own call sequence and buffer pre-fill (rasterize.py:60-90,
    154-190): forward maps + rgb and the three backward kernels against the oracle."""
    import os
    import re
    import types

    from handobjectconsist_amd import _lib

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    code = [b for b in re.findall(r"```python\n(.*?)```", md, re.S) if "def forward_face_index_map" in b]
    assert len(code) == 1
    src = code[0].replace("/path/to/handobjectconsist_amd/libmeshraster_hip.so", _lib.LIB_PATH)
    stub = types.ModuleType("rasterize_stub")
    exec(compile(src, "INTEGRATION.md", "exec"), stub.__dict__)

    B, is_, eps = 2, 64, 1e-3
    faces_np, tex_np = make_case("scene", B, is_, 0)
    saved = R.rasterize_forward(faces_np, tex_np, is_, 0.1, 100, eps, (0, 0, 0), num_threads=8)
    faces, tex = t(faces_np, cuda), t(tex_np, cuda)
    F = faces.shape[1]
    f32 = dict(dtype=torch.float32, device=cuda)
    fim = torch.full((B, is_, is_), -1, dtype=torch.int32, device=cuda)       # rasterize.py:60-85
    wmap = torch.zeros((B, is_, is_, 3), **f32)
    dmap = torch.full((B, is_, is_), 100.0, **f32)
    finv_map = torch.zeros((B, is_, is_, 3, 3), **f32)
    faces_inv = torch.zeros_like(faces)
    rgb = torch.zeros((B, is_, is_, 3), **f32)
    sidx = torch.zeros((B, is_, is_, 8), dtype=torch.int32, device=cuda)
    swgt = torch.zeros((B, is_, is_, 8), **f32)
    out = stub.forward_face_index_map(faces, fim, wmap, dmap, finv_map, faces_inv, is_, 0.1, 100, True, True, True)
    assert out[0] is fim and out[3] is finv_map
    stub.forward_texture_sampling(faces, tex, fim, wmap, dmap, rgb, sidx, swgt, is_, eps)
    assert (fim.cpu().numpy() == saved["face_index_map"]).all()
    assert_close(rgb.cpu().numpy(), saved["rgb_map"] * (saved["face_index_map"] >= 0)[..., None], 1e-6, 1e-6, "rgb")
    assert_close(dmap.cpu().numpy(), saved["depth_map"], 1e-6, 0, "depth")
    alpha = (fim >= 0).float()
    rng = np.random.default_rng(3)
    g_rgb = rng.standard_normal((B, is_, is_, 3)).astype(np.float32)
    g_alpha = rng.standard_normal((B, is_, is_)).astype(np.float32)
    g_depth = rng.standard_normal((B, is_, is_)).astype(np.float32)
    gf_ref, gt_ref = R.rasterize_backward(saved, g_rgb, g_alpha, g_depth, num_threads=8)
    grad_faces, grad_tex = torch.zeros_like(faces), torch.zeros_like(tex)          # rasterize.py:154-158
    stub.backward_pixel_map(faces, fim, rgb, alpha, t(g_rgb, cuda), t(g_alpha, cuda), grad_faces, is_, eps, True, True)
    stub.backward_textures(fim, swgt, sidx, t(g_rgb, cuda), grad_tex, F)
    stub.backward_depth_map(faces, dmap, fim, finv_map, wmap, t(g_depth, cuda), grad_faces, is_)
    assert_close(grad_tex.cpu().numpy(), gt_ref, 1e-4, 1e-5 * np.abs(gt_ref).max(), "grad_textures")
    assert_close(grad_faces.cpu().numpy(), gf_ref, 1e-4, 1e-5 * np.abs(gf_ref).max(), "grad_faces")


def test_fastrender_lit_rgba_matches_oracle(cuda):
    """fastrender.render: lit vertex colours, near 0.05 / far 2, non-square crop, background
    compositing -- against the oracle chain (fill-back, nr.lighting, projection, rasterise)."""
    from handobjectconsist_amd.neurender import fastrender

    B, Wd, H = 2, 128, 96
    s = synth.random_scene(B, seed=12, image_size=Wd)
    rng = np.random.default_rng(2)
    cols = rng.uniform(0, 1, (B, s["verts1"].shape[1], 3)).astype(np.float32)
    got = fastrender.render(t(s["verts1"], cuda), t(s["faces"], cuda), (Wd, H), camintrs=t(s["K1"], cuda),
                            colors=t(cols, cuda), bg_color=0.25).cpu().numpy()
    assert got.shape == (B, H, Wd, 4)
    tex = R.batch_vertex_textures(s["faces"], cols)
    f2, tex2 = R.fill_back(s["faces"], tex)
    faces_cam = R.nr_vertices_to_faces(s["verts1"], f2)
    lit = R.nr_lighting(faces_cam, tex2, 0.8, 0.5, (1, 1, 1), (1, 1, 1), (0, 1, 0))
    v = R.nr_projection(s["verts1"], s["K1"], REN_KW["R"], REN_KW["t"], REN_KW["dist_coeffs"], Wd)
    ref = R.rasterize_rgbad(R.nr_vertices_to_faces(v, f2), lit, Wd, False, 0.05, 2, 1e-3, (0, 0, 0), num_threads=8)
    alpha = ref["alpha"][:, :H, :Wd]
    rgb = ref["rgb"][:, :, :H, :Wd] * alpha[:, None] + 0.25 * (1 - alpha[:, None])
    assert_close(got[..., 3], alpha, 0, 0, "alpha")
    assert_close(got[..., :3], rgb.transpose(0, 2, 3, 1), 1e-4, 3e-4, "lit rgb")
    assert 0.02 < alpha.mean() < 0.6


def test_cyclic_face_padding_is_invisible(cuda):
    """the collate pads faces by cyclic repetition; duplicates tie in depth and the lower index
    wins, so maps and images of the padded mesh equal those of the original, face indices included."""
    from handobjectconsist_amd.neurender import rasterize
    from handobjectconsist_amd.utils import collate

    faces, tex = make_case("scene", 2, 96, 7)
    F = faces.shape[1]
    pad = F + 700
    faces_p = np.stack([collate.pad_cyclic(f, pad) for f in faces])
    tex_p = np.stack([collate.pad_cyclic(x, pad) for x in tex])
    a = rasterize.rasterize_rgbad(t(faces, cuda), t(tex, cuda), 96, False, 0.1, 100, 1e-3, (0, 0, 0))
    b = rasterize.rasterize_rgbad(t(faces_p, cuda), t(tex_p, cuda), 96, False, 0.1, 100, 1e-3, (0, 0, 0))
    for k in ("rgb", "alpha", "depth", "weight_map", "face_index_map"):
        assert torch.equal(a[k], b[k]), k
    assert int(b["face_index_map"].max()) < F


@pytest.mark.parametrize("B,is_,outs", [(3, 96, (1, 1, 1)), (2, 256, (1, 1, 1)), (5, 64, (1, 0, 1)), (2, 128, (0, 1, 0))])
def test_dense_forwards_over_the_tile_list_equal_one_workgroup_per_tile(cuda, B, is_, outs):
    """mr_render_forward / mr_render_vc_forward launch the tiles that hold candidates from the binning pass's list and
    stream the background of all others with a separate kernel (round 4); MR_FLAG_TILE_PER_WORKGROUP keeps the launch of
    rounds 1-3.  Every output byte must be the same, on poisoned buffers (every pixel of every requested plane is
    written by exactly one of the two kernels), for every combination of requested planes."""
    from handobjectconsist_amd import _lib
    from handobjectconsist_amd.neurender import nr_ops
    from handobjectconsist_amd.utils import textutils

    d = _vc_abi_case(cuda, B, is_, 41)
    P, st = _lib.ptr, _lib.stream_ptr(cuda)
    f32 = dict(dtype=torch.float32, device=cuda)
    V, F0 = d["V"], d["F0"]
    bg = torch.tensor([0.25, -0.5, 0.75], **f32)
    rr, ra, rd = outs
    # materialised faces / textures of the same scene for the generic entry point (fill-back by concatenation)
    fidx64 = d["fidx"].long()
    tex = textutils.batch_vertex_textures(fidx64, d["cols"])
    faces = nr_ops.vertices_to_faces(d["v"], torch.cat((fidx64, fidx64.flip(-1)), 1)).contiguous()
    tex2 = torch.cat((tex, tex.permute(0, 1, 4, 3, 2, 5)), 1).contiguous()
    F = faces.shape[1]
    wbytes = int(_lib.load().mr_render_workspace_bytes(B, F, is_))

    def planes():
        return (torch.full((B, 3, is_, is_), float("nan"), **f32), torch.full((B, is_, is_), float("nan"), **f32),
                torch.full((B, is_, is_), float("nan"), **f32), torch.full((B, is_, is_), -7, dtype=torch.int32, device=cuda),
                torch.full((B, is_, is_, 3), float("nan"), **f32))

    def run(vc, flags):
        rgb, alpha, depth, fim, wmap = planes()
        work = torch.full((wbytes,), 0xAB, dtype=torch.uint8, device=cuda)
        if vc:
            _lib.call("mr_render_vc_forward", P(d["v"]), P(d["fidx"]), P(d["cols"]), P(bg), 0, P(rgb), P(alpha), P(depth), P(fim),
                      P(wmap), P(work), wbytes, B, V, F0, 1, is_, 0.1, 100.0, 1e-3, rr, ra, rd, flags, 0, st)
        else:
            _lib.call("mr_render_forward", P(faces), P(tex2), P(bg), 0, P(rgb), P(alpha), P(depth), P(fim), P(wmap), None, P(work),
                      wbytes, B, F, is_, 2, 0.1, 100.0, 1e-3, rr, ra, rd, flags, st)
        torch.cuda.synchronize()
        return [x.view(torch.int32) for x in (rgb, alpha, depth, fim, wmap)]

    for vc in (True, False):
        old = run(vc, _lib.FLAG_TILE_PER_WORKGROUP)
        new = run(vc, 0)
        for a, b_, name, wanted in zip(old, new, ("rgb", "alpha", "depth", "face_index_map", "weight_map"), (rr, ra, rd, 1, 1)):
            assert torch.equal(a, b_), f"{name} differs (vertex colours: {vc})"
            if wanted:  # ... and nothing of a requested plane kept its poison
                poison = -7 if name == "face_index_map" else torch.tensor(float("nan")).view(torch.int32).item()
                assert int((b_ == poison).sum()) == 0, f"{name}: unwritten pixels (vertex colours: {vc})"
        assert int((new[3] >= 0).sum()) > 100


FLAG_ONE_WORKGROUP_PER_IMAGE = 32 << 24  # (profiling / A-B bit of the binning pass: raster_fwd.hip, launch_bins)
FLAG_FORCE_PARTS = 1 << 24  # (launch_bins: several workgroups per image also where one is the default)
FLAG_PARTS_IN_ONE_LAUNCH = 64 << 24  # (launch_bins: the last-arriver form also where four or more parts take two launches, round 6)


@pytest.mark.parametrize("B,is_", [(2, 256), (13, 96), (40, 64), (3, 480)])
def test_binning_pass_in_parts_equals_one_workgroup_per_image(cuda, B, is_):
    """Round 5: an image is binned by several workgroups (a contiguous range of its faces each; they exchange their bin
    counters behind a barrier on the image's arrival counter).  Every output byte of the listed flow-mode render, of the
    dense vertex-colour render and of the generic render and the reported tile-list length equal the one-workgroup-per-image pass, on workspaces full of 0xff (the arrival counters must come from
    the per-face pass, not from the allocation); batch sizes that are no multiple of 8 leave surplus workgroups."""
    from handobjectconsist_amd import _lib
    from handobjectconsist_amd.neurender import nr_ops
    from handobjectconsist_amd.utils import textutils

    d = _vc_abi_case(cuda, B, is_, 57)
    P, st = _lib.ptr, _lib.stream_ptr(cuda)
    f32 = dict(dtype=torch.float32, device=cuda)
    V, F0 = d["V"], d["F0"]
    assert 2 * F0 >= 512  # (enough faces for more than one part)
    bg = torch.tensor([0.25, -0.5, 0.75], **f32)
    wbytes = int(_lib.load().mr_render_workspace_bytes(B, 2 * F0, is_))
    word = torch.zeros(1, dtype=torch.int32).pin_memory()
    fidx64 = d["fidx"].long()
    tex = textutils.batch_vertex_textures(fidx64, d["cols"])
    faces = nr_ops.vertices_to_faces(d["v"], torch.cat((fidx64, fidx64.flip(-1)), 1)).contiguous()
    tex2 = torch.cat((tex, tex.permute(0, 1, 4, 3, 2, 5)), 1).contiguous()

    def run(kind, flags):
        rgb = torch.full((B, 3, is_, is_), float("nan"), **f32)
        alpha, mask, depth = (torch.full((B, is_, is_), float("nan"), **f32) for _ in range(3))
        wmap = torch.full((B, is_, is_, 3), float("nan"), **f32)
        fim = torch.full((B, is_, is_), -7, dtype=torch.int32, device=cuda)
        vid = torch.full((B, is_, is_, 3), -7, dtype=torch.int32, device=cuda)
        hit = torch.full((B, (is_ + 7) // 8, (is_ + 31) // 32, 4), 9, dtype=torch.uint8, device=cuda)
        work = torch.full((wbytes,), 0xFF, dtype=torch.uint8, device=cuda)
        word[0] = -1
        if kind == "flow":
            _lib.call("mr_render_flow_forward", P(d["v"]), P(d["fidx"]), P(d["cols"]), P(bg), 0, None, 0, 0.99999, P(rgb), P(alpha),
                      P(mask), None, P(wmap), P(fim), P(hit), P(work), wbytes, B, V, F0, 1, is_, 0.1, 100.0, 1e-3,
                      _lib.FLAG_SPARSE_TILES | flags, P(vid), -1, P(word), None, 0, 0, st)
        elif kind == "vc":
            _lib.call("mr_render_vc_forward", P(d["v"]), P(d["fidx"]), P(d["cols"]), P(bg), 0, P(rgb), P(alpha), P(depth), P(fim),
                      P(wmap), P(work), wbytes, B, V, F0, 1, is_, 0.1, 100.0, 1e-3, 1, 1, 1, flags, 0, st)
        else:
            _lib.call("mr_render_forward", P(faces), P(tex2), P(bg), 0, P(rgb), P(alpha), P(depth), P(fim), P(wmap), None, P(work),
                      wbytes, B, 2 * F0, is_, 2, 0.1, 100.0, 1e-3, 1, 1, 1, flags, st)
        torch.cuda.synchronize()
        outs = [x.view(torch.int32) if x.dtype == torch.float32 else x for x in (rgb, alpha, mask, depth, wmap, fim, vid, hit)]
        return outs, int(word[0])

    for kind in ("flow", "vc", "generic"):
        one, n_one = run(kind, FLAG_ONE_WORKGROUP_PER_IMAGE)
        # (FLAG_FORCE_PARTS: since round 6 the standalone entry points bin rasters of up to 1024 tiles with one workgroup per
        # image -- the parts only pay where they split the per-face pass, i.e. on the training path's fused launch, which
        # tests/test_gpu_warp.py drives; the switch keeps this comparison on the parts)
        # (round 6: four or more parts exchange their counters across a kernel boundary -- count launch, fill launch; fewer, or
        # FLAG_PARTS_IN_ONE_LAUNCH, through the memory side with the last arriver finishing the image: both forms here)
        for flags in (FLAG_FORCE_PARTS, FLAG_FORCE_PARTS, FLAG_FORCE_PARTS | FLAG_PARTS_IN_ONE_LAUNCH, 0):  # (twice: the second call finds the first one's counters in a REUSED allocation)
            parts, n_parts = run(kind, flags)
            for a, b_, name in zip(one, parts, ("rgb", "alpha", "mask", "depth", "weights", "face_index_map", "vertex ids", "coverage")):
                assert torch.equal(a, b_), f"{name} differs ({kind})"
            assert n_one == n_parts
        assert int((one[5] >= 0).sum()) > 100
        if kind == "flow":
            assert n_one > 8
