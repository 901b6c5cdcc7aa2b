"""Generate tests/golden/chain_*.npz by RUNNING THE REFERENCE'S OWN GLUE on CPU.

Build container only (needs /root/reference).  The committed .npz files are data: seeded inputs and
what the reference's code returned for them -- nothing of the reference's source.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_chain.py

What executes, unmodified, from /root/reference (see oracle/ref_glue.py for the import recipe):
meshreg/neurender/rasterize.py (RasterizeFunction forward + backward, rasterize_rgbad and wrappers),
meshreg/neurender/renderer.py (Renderer.*), meshreg/warping/opticalflow.py, meshreg/warping/imgflowarp.py,
meshreg/optim/pyramidloss.py, meshreg/models/warpbranch.py.  The third-party packages they call and
which are absent from the image are stubbed: the five ``neural_renderer.cuda.rasterize`` entry points
by the C oracle (oracle/raster_oracle.c -- PARITY UNPINNED, source absent), the ``neural_renderer`` /
``libyana`` python helpers by torch restatements of SURVEY appendix B.  So these fixtures pin every
line of the reference's Python on the path -- buffer pre-fills, background / alpha, NHWC->NCHW, the
vertical flip and which maps are NOT flipped, anti-aliasing, eps per entry point, fill-back, the
mask algebra of get_opticalflow with its quirks, crop, GT-reference substitution, detach of
frames > 0, stack().mean() -- and the autograd chain through it down to the mesh vertices; they do
not pin the six kernels themselves.

Files:
    chain_rasterize.npz    rasterize_rgbad: AA on/off x every return_* combination x tuple / [B,3]
                           background, ts = 2 and 3; outputs + grad_faces + grad_textures
    chain_renderer.npz     Renderer.render (projection camera, per-sample K, distortion, fill-back,
                           lighting, detach_renders), render_rgb / _silhouettes / _depth, project, look_at
    chain_opticalflow.npz  get_opticalflow: ignore list, non-square crop, detach_* and mask_occlusions
                           combinations; flows + d/d vertices of both frames
    chain_opticalflow_cfg.npz  get_opticalflow (training setting) at the raster sizes of BASELINE.json's configs:
                           480 (crop 480 x 270) and 640 (crop 640 x 480); d/d vertices in full, flows as seeded samples
    chain_warpbranch.npz   warpbranch.forward: gt_refs, use_backward, first_only, 2 and 3 frames;
                           loss, per-pair losses, flows, masks, d loss / d predicted vertices
"""
import itertools
import json
import os
import sys
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_glue  # noqa: E402

ref = ref_glue.install()
OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(4)
torch.manual_seed(0)


def T(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.requires_grad_(True) if grad else t


def N(t):
    return None if t is None else t.detach().cpu().numpy().copy()


def save(name, arrays, meta):
    arrays = {k: v for k, v in arrays.items() if v is not None}
    arrays["meta"] = np.array(json.dumps(meta))
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {len(arrays)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


# ---------------------------------------------------------------------------------------------------
# meshes
# ---------------------------------------------------------------------------------------------------


def icosphere(subdiv):
    """Closed, outward-oriented (counter-clockwise seen from outside) triangle mesh of the unit sphere."""
    p = (1 + 5 ** 0.5) / 2
    v = [(-1, p, 0), (1, p, 0), (-1, -p, 0), (1, -p, 0), (0, -1, p), (0, 1, p), (0, -1, -p), (0, 1, -p),
         (p, 0, -1), (p, 0, 1), (-p, 0, -1), (-p, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2),
         (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11),
         (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.asarray(x, np.float64) / np.linalg.norm(x) for x in v]
    for _ in range(subdiv):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.asarray(v, np.float32), np.asarray(f, np.int64)


def blob(rng, subdiv, radii, centre, wobble=0.15):
    v, f = icosphere(subdiv)
    v = v * (1 + wobble * rng.standard_normal((v.shape[0], 1)).astype(np.float32))
    rot, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    v = (v * np.asarray(radii, np.float32)) @ rot.astype(np.float32).T + np.asarray(centre, np.float32)
    return v.astype(np.float32), f


def scene(rng, B, image_size, hand_subdiv=2, obj_subdiv=1, motion=0.004):
    """Hand-like + object-like blobs about half a metre from a pinhole camera whose principal point sits
    near the image centre; frame 2 = frame 1 moved by a few millimetres (flows of a few pixels)."""
    hv, ov, hv2, ov2, K1, K2 = [], [], [], [], [], []
    for _ in range(B):
        z = rng.uniform(0.4, 0.6)
        c = np.array([rng.uniform(-0.03, 0.03), rng.uniform(-0.03, 0.03), z])
        h, hf = blob(rng, hand_subdiv, (0.05, 0.08, 0.03), c)
        o, of = blob(rng, obj_subdiv, (0.04, 0.04, 0.06), c + rng.uniform(-0.05, 0.05, 3) * (1, 1, 0.3))
        hv.append(h), ov.append(o)
        d = rng.standard_normal(3) * motion
        hv2.append(h + d.astype(np.float32) + (rng.standard_normal(h.shape) * motion * 0.3).astype(np.float32))
        ov2.append(o + (rng.standard_normal(3) * motion).astype(np.float32))
        for Ks in (K1, K2):
            f = rng.uniform(1.2, 1.6) * image_size
            Ks.append(np.array([[f, 0, image_size / 2 + rng.uniform(-2, 2)], [0, f, image_size / 2 + rng.uniform(-2, 2)],
                                [0, 0, 1]], np.float32))
    st = lambda x: np.stack(x).astype(np.float32)
    return dict(hand1=st(hv), obj1=st(ov), hand2=st(hv2), obj2=st(ov2), hand_faces=hf, obj_faces=of, K1=st(K1),
                K2=st(K2))


def smooth_images(rng, B, H, W, amp=0.5):
    """Low-frequency random images in [-0.5, 0.5] (a photometric loss needs gradients that mean something)."""
    coarse = torch.from_numpy(rng.uniform(-1, 1, (B, 3, 7, 9)).astype(np.float32))
    img = torch.nn.functional.interpolate(coarse, size=(H, W), mode="bicubic", align_corners=True)
    img = img * amp + torch.from_numpy(rng.uniform(-0.03, 0.03, (B, 3, H, W)).astype(np.float32))
    return img.clamp(-0.5, 0.5).numpy().astype(np.float32)


def jitter_mask(rng, B, H, W, maxb=3):
    m = np.ones((B, 3, H, W), np.float32)
    for b in range(B):
        l, r, u, d = rng.integers(0, maxb + 1, size=4)
        if l: m[b, :, :, :l] = 0
        if r: m[b, :, :, W - r:] = 0
        if u: m[b, :, :u, :] = 0
        if d: m[b, :, H - d:, :] = 0
    return m


# ---------------------------------------------------------------------------------------------------
# 1. rasterize_rgbad
# ---------------------------------------------------------------------------------------------------


def random_faces(rng, B, F, tiny=4, back=4):
    f = rng.uniform(-1.1, 1.1, (B, F, 3, 3)).astype(np.float32)
    ctr = rng.uniform(-0.8, 0.8, (B, F, 1, 2)).astype(np.float32)
    f[:, :, :, :2] = ctr + rng.uniform(-0.45, 0.45, (B, F, 3, 2)).astype(np.float32)
    f[:, :tiny, :, :2] = ctr[:, :tiny] + rng.uniform(-0.05, 0.05, (B, tiny, 3, 2)).astype(np.float32)
    f[:, :, :, 2] = rng.uniform(0.3, 3.0, (B, F, 3)).astype(np.float32)
    f[:, -2:, :, 2] = rng.uniform(0.02, 0.2, (B, 2, 3))  # partly in front of the near plane
    f[0, 5] = f[0, 6]  # coplanar duplicate: lowest index wins
    return f


def gen_rasterize():
    rng = np.random.default_rng(11)
    B, F, is_ = 2, 28, 12
    arrays, meta = {}, []
    faces = random_faces(rng, B, F)
    arrays["faces"] = faces
    for ts in (2, 3):
        arrays[f"textures_ts{ts}"] = rng.uniform(0, 1, (B, F, ts, ts, ts, 3)).astype(np.float32)
    for aa in (False, True):
        s = is_
        arrays[f"g_rgb_{s}"] = rng.standard_normal((B, 3, s, s)).astype(np.float32)
        arrays[f"g_alpha_{s}"] = rng.standard_normal((B, s, s)).astype(np.float32)
        arrays[f"g_depth_{s}"] = rng.standard_normal((B, s, s)).astype(np.float32)
    backgrounds = {"tuple": (0.2, 0.4, 0.6), "per_sample": [[0.1, 0.2, 0.3], [0.9, 0.5, 0.0]]}
    combos = [c for c in itertools.product((False, True), repeat=3) if any(c)]
    idx = 0
    for aa, (rr, ra, rd), bgname, ts in itertools.product((False, True), combos, backgrounds, (2, 3)):
        if ts == 3 and not (rr and ra and rd):
            continue
        if bgname == "per_sample" and not rr:
            continue
        eps = 1e-3 if idx % 2 == 0 else 1e-4
        ft = T(faces, True)
        tt = T(arrays[f"textures_ts{ts}"], True) if rr else None
        out = ref.rasterize.rasterize_rgbad(ft, tt, is_, aa, 0.1, 100, eps, backgrounds[bgname], rr, ra, rd)
        loss = 0
        if rr:
            loss = loss + (out["rgb"] * T(arrays[f"g_rgb_{is_}"])).sum()
        if ra:
            loss = loss + (out["alpha"] * T(arrays[f"g_alpha_{is_}"])).sum()
        if rd:
            loss = loss + (out["depth"] * T(arrays[f"g_depth_{is_}"])).sum()
        loss.backward()
        key = f"c{idx}"
        for name in ("rgb", "alpha", "depth", "face_index_map", "weight_map", "face_inv_map"):
            arrays[f"{key}_{name}"] = N(out[name])
        arrays[f"{key}_grad_faces"] = N(ft.grad)
        arrays[f"{key}_grad_textures"] = N(tt.grad) if rr else None
        meta.append(dict(key=key, anti_aliasing=aa, return_rgb=rr, return_alpha=ra, return_depth=rd,
                         background=bgname, background_value=backgrounds[bgname], ts=ts, eps=eps, image_size=is_,
                         near=0.1, far=100))
        idx += 1
    # the thin wrappers (rasterize.py:451-536) with their own defaults
    ft = T(faces)
    arrays["w_rasterize"] = N(ref.rasterize.rasterize(ft, T(arrays["textures_ts2"]), is_))
    arrays["w_silhouettes"] = N(ref.rasterize.rasterize_silhouettes(ft, is_))
    arrays["w_depth"] = N(ref.rasterize.rasterize_depth(ft, is_, False))
    save("chain_rasterize.npz", arrays, meta)


# ---------------------------------------------------------------------------------------------------
# 2. Renderer
# ---------------------------------------------------------------------------------------------------


def gen_renderer():
    rng = np.random.default_rng(12)
    B, is_ = 2, 24
    sc = scene(rng, B, is_, hand_subdiv=1, obj_subdiv=0)
    verts = np.concatenate([sc["hand1"], sc["obj1"]], 1)
    faces = np.concatenate([sc["hand_faces"], sc["obj_faces"] + sc["hand1"].shape[1]], 0)[None].repeat(B, 0)
    # a few faces with reversed winding so that fill_back matters
    faces[:, ::7] = faces[:, ::7, ::-1]
    F0 = faces.shape[1]
    tex = rng.uniform(0, 1, (B, F0, 2, 2, 2, 3)).astype(np.float32)
    K = sc["K1"]
    R = np.eye(3, dtype=np.float32)[None]
    Rrot = np.array([[[0.9950042, -0.0998334, 0], [0.0998334, 0.9950042, 0], [0, 0, 1]]], np.float32)
    t0 = np.zeros((1, 3), np.float32)
    t1 = np.array([[0.01, -0.02, 0.03]], np.float32)
    dist = np.array([[0.1, -0.05, 0.002, -0.001, 0.01]], np.float32)
    arrays = dict(verts=verts, faces=faces, textures=tex, K=K, R_eye=R, R_rot=Rrot, t_zero=t0, t_off=t1, dist=dist)
    arrays["g_rgb"] = rng.standard_normal((B, 3, is_, is_)).astype(np.float32)
    arrays["g_alpha"] = rng.standard_normal((B, is_, is_)).astype(np.float32)
    arrays["g_depth"] = rng.standard_normal((B, is_, is_)).astype(np.float32)
    meta = []

    def run_render(key, ctor, call, detach_renders=False):
        ren = ref.renderer.Renderer(**{k: (T(arrays[v]) if isinstance(v, str) else v) for k, v in ctor.items()})
        vt, tt = T(verts, True), T(tex, True)
        # the projected vertices the reference's render fed its rasteriser with, and the gradient that came back to
        # them: a test that hands THESE to the HIP rasteriser sees no difference between two fp32 projections
        nr_mod, seen = sys.modules["neural_renderer"], []
        real_projection = nr_mod.projection

        def recording_projection(*a, **kw):
            out = real_projection(*a, **kw)
            if out.requires_grad:
                out.retain_grad()
            seen.append(out)
            return out

        nr_mod.projection = recording_projection
        try:
            out = ren(vt, T(faces), tt, detach_renders=detach_renders,
                      **{k: (T(arrays[v]) if isinstance(v, str) else v) for k, v in call.items()})
        finally:
            nr_mod.projection = real_projection
        loss = (out["rgb"] * T(arrays["g_rgb"])).sum() + (out["alpha"] * T(arrays["g_alpha"])).sum() \
            + (out["depth"] * T(arrays["g_depth"])).sum()
        loss.backward()
        assert len(seen) == 1
        arrays[f"{key}_ndc"] = N(seen[0])
        arrays[f"{key}_grad_ndc"] = N(seen[0].grad) if seen[0].grad is not None else None
        for name in ("rgb", "alpha", "depth", "face_index_map", "weight_map", "face_inv_map"):
            arrays[f"{key}_{name}"] = N(out[name])
        arrays[f"{key}_grad_verts"] = N(vt.grad)
        arrays[f"{key}_grad_textures"] = N(tt.grad)
        meta.append(dict(key=key, kind="render", ctor=ctor, call=call, detach_renders=detach_renders))

    train = dict(image_size=is_, R="R_eye", t="t_zero", K="K", orig_size=is_, anti_aliasing=False, fill_back=True,
                 near=0.1, no_light=True)  # warpreg.py:40-51
    run_render("train", train, {}, detach_renders=True)
    run_render("train_attached", train, {}, detach_renders=False)
    run_render("call_K", dict(train, K=None), {"K": "K"}, detach_renders=True)  # opticalflow.py:108 passes K per call
    run_render("pose_dist", dict(train, R="R_rot", t="t_off", dist_coeffs="dist"), {}, detach_renders=False)
    run_render("no_fill_back_aa", dict(train, fill_back=False, anti_aliasing=True,
                                       background_color=[0.3, 0.1, 0.7]), {})
    vis = dict(image_size=is_, R="R_eye", t="t_zero", K="K", orig_size=is_, anti_aliasing=False, fill_back=True,
               near=0.05, far=2, no_light=False, light_intensity_ambient=0.8)  # fastrender.py:33-46
    run_render("lit", vis, {})
    run_render("lit_dir", dict(vis, light_intensity_directional=0.7, light_color_directional=[1.0, 0.5, 0.25],
                               light_direction=[0.0, 0.6, -0.8], light_color_ambient=[0.5, 1.0, 0.75]), {})

    def run_mode(key, ctor, mode):
        ren = ref.renderer.Renderer(**{k: (T(arrays[v]) if isinstance(v, str) else v) for k, v in ctor.items()})
        vt, tt = T(verts, True), T(tex, True)
        if mode == "project":
            out = ren.project(vt)
            g = torch.ones_like(out) * torch.tensor([1.0, -2.0, 0.5])
        else:
            out = ren(vt, T(faces), tt, mode=mode)
            g = T(arrays["g_rgb"]) if mode == "rgb" else T(arrays["g_alpha"])
        (out * g).sum().backward()
        arrays[f"{key}_out"] = N(out)
        arrays[f"{key}_grad_verts"] = N(vt.grad)
        arrays[f"{key}_grad_textures"] = N(tt.grad) if tt.grad is not None else None
        meta.append(dict(key=key, kind=mode, ctor=ctor))

    run_mode("m_rgb", train, "rgb")
    run_mode("m_sil", train, "silhouettes")
    run_mode("m_sil_aa", dict(train, anti_aliasing=True), "silhouettes")
    run_mode("m_depth", train, "depth")
    run_mode("m_depth_aa", dict(train, anti_aliasing=True, fill_back=False), "depth")
    run_mode("m_project", dict(train, R="R_rot", t="t_off", dist_coeffs="dist"), "project")
    # look_at camera on an object-centred mesh (API parity; no caller on the training path)
    unit = (verts - verts.mean(1, keepdims=True)) * 6
    arrays["verts_unit"] = unit.astype(np.float32)
    for key, ctor in (("look_at", dict(image_size=is_, camera_mode="look_at", anti_aliasing=False, no_light=True)),
                      ("look", dict(image_size=is_, camera_mode="look", anti_aliasing=True, viewing_angle=25))):
        ren = ref.renderer.Renderer(**ctor)
        vt, tt = T(arrays["verts_unit"], True), T(tex, True)
        out = ren(vt, T(faces), tt)
        ((out["rgb"] * T(arrays["g_rgb"])).sum() + (out["alpha"] * T(arrays["g_alpha"])).sum()).backward()
        for name in ("rgb", "alpha", "depth", "face_index_map"):
            arrays[f"{key}_{name}"] = N(out[name])
        arrays[f"{key}_grad_verts"] = N(vt.grad)
        arrays[f"{key}_grad_textures"] = N(tt.grad)
        meta.append(dict(key=key, kind="render_unit", ctor=ctor))
    save("chain_renderer.npz", arrays, meta)


# ---------------------------------------------------------------------------------------------------
# 3. get_opticalflow
# ---------------------------------------------------------------------------------------------------


def training_renderer(is_):
    """The instance WarpRegNet builds (warpreg.py:40-51)."""
    return ref.renderer.Renderer(image_size=is_, R=torch.eye(3).unsqueeze(0), t=torch.zeros(1, 3),
                                 K=torch.ones(1, 3, 3), orig_size=is_, anti_aliasing=False, fill_back=True,
                                 near=0.1, no_light=True)


def gen_opticalflow():
    rng = np.random.default_rng(13)
    arrays, meta = {}, []
    for sname, (B, is_, crop) in {"sq": (2, 40, None), "crop": (2, 48, (48, 27)), "one": (1, 64, (64, 36))}.items():
        sc = scene(rng, B, is_)
        Vh, Fh = sc["hand1"].shape[1], sc["hand_faces"].shape[0]
        v1 = np.concatenate([sc["hand1"], sc["obj1"]], 1)
        v2 = np.concatenate([sc["hand2"], sc["obj2"]], 1)
        faces = np.concatenate([sc["hand_faces"], sc["obj_faces"] + Vh], 0)[None].repeat(B, 0)
        ignore = list(range(Fh - 24, Fh))  # the LAST hand faces, like the 14 wrist-closing faces (manoutils.py:33)
        H, W = (crop[1], crop[0]) if crop else (is_, is_)
        arrays.update({f"{sname}_verts1": v1, f"{sname}_verts2": v2, f"{sname}_faces": faces, f"{sname}_K1": sc["K1"],
                       f"{sname}_K2": sc["K2"],
                       f"{sname}_g12": rng.standard_normal((B, H, W, 2)).astype(np.float32),
                       f"{sname}_g21": rng.standard_normal((B, H, W, 2)).astype(np.float32)})
        variants = [dict(ignore=True, detach_textures=False, detach_renders=True, mask_occlusions=True)]
        if sname == "sq":
            variants += [dict(ignore=False, detach_textures=False, detach_renders=True, mask_occlusions=True),
                         dict(ignore=True, detach_textures=True, detach_renders=True, mask_occlusions=True),
                         dict(ignore=True, detach_textures=False, detach_renders=False, mask_occlusions=True),
                         dict(ignore=True, detach_textures=False, detach_renders=True, mask_occlusions=False)]
        for vi, var in enumerate(variants):
            a, b = T(v1, True), T(v2, True)
            flows = ref.opticalflow.get_opticalflow(
                [a, b], T(faces), [T(sc["K1"]), T(sc["K2"])], training_renderer(is_), orig_img_size=crop,
                mask_occlusions=var["mask_occlusions"], detach_textures=var["detach_textures"],
                detach_renders=var["detach_renders"], ignore_face_idxs=ignore if var["ignore"] else None)
            loss = (flows[0] * T(arrays[f"{sname}_g12"])).sum() + (flows[1] * T(arrays[f"{sname}_g21"])).sum()
            loss.backward()
            key = f"{sname}_v{vi}"
            arrays[f"{key}_flow12"], arrays[f"{key}_flow21"] = N(flows[0]), N(flows[1])
            arrays[f"{key}_grad_verts1"] = N(a.grad) if a.grad is not None else np.zeros_like(v1)
            arrays[f"{key}_grad_verts2"] = N(b.grad) if b.grad is not None else np.zeros_like(v2)
            meta.append(dict(key=key, scene=sname, image_size=is_, orig_img_size=crop, ignore_face_idxs=ignore,
                             **var))
            print(key, "covered px:", int((flows[0][..., 0] != 0).sum()), int((flows[1][..., 0] != 0).sum()),
                  "|grad1|", float(a.grad.abs().sum()) if a.grad is not None else 0.0)
    save("chain_opticalflow.npz", arrays, meta)


def flow_grad_inputs(seed, B, H, W):
    """The upstream gradients of the two flows, reproducible from a seed (tests regenerate them instead of reading
    megabytes of noise from the fixture)."""
    r = np.random.default_rng(seed)
    return r.standard_normal((B, H, W, 2)).astype(np.float32), r.standard_normal((B, H, W, 2)).astype(np.float32)


def gen_opticalflow_config_sizes():
    """get_opticalflow at the raster sizes of BASELINE.json's configs 2 and 4 (480 x 270 frame pairs; 640 x 480
    frames), training setting.  Stored: the inputs, d loss / d vertices in full, and of the two flows a seeded sample of
    40 000 pixels each plus support counts and sums (the flows themselves are megabytes)."""
    rng = np.random.default_rng(17)
    arrays, meta = {}, []
    for sname, (B, is_, crop, gseed) in {"c480": (1, 480, (480, 270), 101), "c640": (1, 640, (640, 480), 102)}.items():
        sc = scene(rng, B, is_, hand_subdiv=3, obj_subdiv=2)
        Vh, Fh = sc["hand1"].shape[1], sc["hand_faces"].shape[0]
        v1 = np.concatenate([sc["hand1"], sc["obj1"]], 1)
        v2 = np.concatenate([sc["hand2"], sc["obj2"]], 1)
        faces = np.concatenate([sc["hand_faces"], sc["obj_faces"] + Vh], 0)[None].repeat(B, 0)
        ignore = list(range(Fh - 24, Fh))
        H, W = crop[1], crop[0]
        g12, g21 = flow_grad_inputs(gseed, B, H, W)
        a, b = T(v1, True), T(v2, True)
        flows = ref.opticalflow.get_opticalflow([a, b], T(faces), [T(sc["K1"]), T(sc["K2"])], training_renderer(is_),
                                                orig_img_size=crop, mask_occlusions=True, detach_textures=False,
                                                detach_renders=True, ignore_face_idxs=ignore)
        loss = (flows[0] * T(g12)).sum() + (flows[1] * T(g21)).sum()
        loss.backward()
        idx = np.random.default_rng(gseed + 1000).choice(B * H * W, 40000, replace=False)
        arrays.update({f"{sname}_verts1": v1, f"{sname}_verts2": v2, f"{sname}_faces": faces, f"{sname}_K1": sc["K1"],
                       f"{sname}_K2": sc["K2"], f"{sname}_grad_verts1": N(a.grad), f"{sname}_grad_verts2": N(b.grad),
                       f"{sname}_sample_idx": idx.astype(np.int64)})
        for i, name in enumerate(("flow12", "flow21")):
            fl = N(flows[i]).reshape(-1, 2)
            arrays[f"{sname}_{name}_sample"] = fl[idx]
            arrays[f"{sname}_{name}_support"] = np.array([(fl[:, 0] != 0).sum(), (fl[:, 1] != 0).sum()], np.int64)
            arrays[f"{sname}_{name}_sum"] = fl.astype(np.float64).sum(0)
        meta.append(dict(key=sname, scene=sname, image_size=is_, orig_img_size=crop, ignore_face_idxs=ignore,
                         grad_seed=gseed, batch=B))
        print(sname, "faces", faces.shape[1], "covered px", int((flows[0][..., 0] != 0).sum()),
              int((flows[1][..., 0] != 0).sum()), "|grad1|", float(a.grad.abs().sum()), "loss", float(loss))
    save("chain_opticalflow_cfg.npz", arrays, meta)


# ---------------------------------------------------------------------------------------------------
# 4. warpbranch.forward
# ---------------------------------------------------------------------------------------------------


def gen_warpbranch():
    rng = np.random.default_rng(14)
    TQ, BQ = ref.queries.TransQueries, ref.queries.BaseQueries
    B, is_, crop = 2, 48, (48, 32)
    H, W = crop[1], crop[0]
    sc = scene(rng, B, is_)
    sc3 = scene(rng, B, is_)  # an unrelated third frame is fine: only frame 0 <-> frame k pairs are formed
    Fh = sc["hand_faces"].shape[0]
    ignore = list(range(Fh - 24, Fh))
    arrays, meta = {}, []
    frames = []
    for k in range(3):
        src = sc if k < 2 else sc3
        suffix = "1" if k == 0 else "2"
        # network prediction for this frame and (for k > 0) its ground truth, deliberately different
        pred_hand, pred_obj = src["hand" + suffix], src["obj" + suffix]
        if k == 2:  # third frame: frame 0's geometry moved a little more
            pred_hand = sc["hand1"] + np.float32(0.006) * rng.standard_normal((B, 1, 3)).astype(np.float32)
            pred_obj = sc["obj1"] + np.float32(0.006) * rng.standard_normal((B, 1, 3)).astype(np.float32)
        gt_hand = pred_hand + (rng.standard_normal((B, 1, 3)) * 0.003).astype(np.float32)
        gt_obj = pred_obj + (rng.standard_normal((B, 1, 3)) * 0.003).astype(np.float32)
        fr = dict(image=smooth_images(rng, B, H, W), jittermask=jitter_mask(rng, B, H, W),
                  camintr=(sc["K1"] if k == 0 else sc["K2"]), objfaces=sc["obj_faces"][None].repeat(B, 0),
                  pred_hand=pred_hand.astype(np.float32), pred_obj=pred_obj.astype(np.float32),
                  gt_hand=gt_hand.astype(np.float32), gt_obj=gt_obj.astype(np.float32))
        frames.append(fr)
        for name, val in fr.items():
            arrays[f"f{k}_{name}"] = val
    arrays["hand_face"] = sc["hand_faces"]
    criterion = ref.pyramidloss.PyramidCriterion("l1")
    cases = [dict(frames=2, gt_refs=True, use_backward=True, first_only=True),
             dict(frames=2, gt_refs=False, use_backward=True, first_only=True),
             dict(frames=2, gt_refs=True, use_backward=False, first_only=True),
             dict(frames=2, gt_refs=False, use_backward=True, first_only=False),
             dict(frames=3, gt_refs=True, use_backward=True, first_only=True)]
    for ci, case in enumerate(cases):
        samples, results = [], []
        for fr in frames[: case["frames"]]:
            samples.append({TQ.IMAGE: T(fr["image"]), TQ.JITTERMASK: T(fr["jittermask"]), TQ.CAMINTR: T(fr["camintr"]),
                            BQ.OBJFACES: T(fr["objfaces"]), BQ.OBJVERTS3D: T(fr["gt_obj"]),
                            BQ.HANDVERTS3D: T(fr["gt_hand"])})
            results.append({"recov_handverts3d": T(fr["pred_hand"], True), "recov_objverts3d": T(fr["pred_obj"], True)})
        loss, pair = ref.warpbranch.forward(
            samples, results, T(sc["hand_faces"])[None], training_renderer(is_), crop, criterion,
            gt_refs=case["gt_refs"], first_only=case["first_only"], hand_ignore_faces=ignore,
            use_backward=case["use_backward"])
        loss.backward()
        key = f"w{ci}"
        arrays[f"{key}_loss"] = N(loss)
        arrays[f"{key}_diff_losses"] = N(pair["diff_losses"])
        for pi, (flows, masks, warps, diffs) in enumerate(zip(pair["recons_flows"], pair["masks"], pair["warps"],
                                                               pair["diffs"])):
            for d in (0, 1):
                arrays[f"{key}_p{pi}_flow{d}"] = N(flows[d])
                arrays[f"{key}_p{pi}_full_mask{d}"] = N(masks[d]["full_mask"])
                arrays[f"{key}_p{pi}_warp_mask{d}"] = N(masks[d]["warp_mask"][:, 0])
                arrays[f"{key}_p{pi}_warp{d}"] = N(warps[d])
                arrays[f"{key}_p{pi}_diff{d}"] = N(diffs[d])
        for k, res in enumerate(results):
            for name in ("recov_handverts3d", "recov_objverts3d"):
                g = res[name].grad
                arrays[f"{key}_f{k}_grad_{name}"] = N(g) if g is not None else np.zeros_like(N(res[name]))
        meta.append(dict(key=key, image_size=is_, input_res=crop, hand_ignore_faces=ignore, **case))
        print(key, "loss", float(loss), "valid px", [int(m[d]["full_mask"].sum()) for m in pair["masks"] for d in (0, 1)],
              "|g hand0|", float(results[0]["recov_handverts3d"].grad.abs().sum()))
    save("chain_warpbranch.npz", arrays, meta)


# ---------------------------------------------------------------------------------------------------
# 5. fastrender.render (visualisation path, SURVEY 8f "f3")
# ---------------------------------------------------------------------------------------------------


def gen_fastrender():
    """meshreg/neurender/fastrender.py:14-59 run from the reference: the lit RGBA render of a posed mesh (ambient 0.8 +
    the renderer's default directional light), crop to the frame, optional background compositing.  Beside the stubs of
    oracle/ref_glue.py the module-level imports of that file need manopth (manoutils) and the figure helpers of
    meshreg.visualize -- none of them is touched by ``render``.  Its compositing line multiplies [B,3,H,W] by
    [B,H,W]: that broadcasts only for B = 1 (and, wrongly, for B = 3: channel c gets the alpha of SAMPLE c); recorded
    here: B = 1 with a background, B = 2 without, and that B = 2 with a background raises."""
    import types

    for name in ("manopth", "manopth.manolayer", "meshreg.visualize", "meshreg.visualize.consistdisplay",
                 "meshreg.visualize.rotateverts", "meshreg.visualize.samplevis"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["manopth.manolayer"].ManoLayer = object
    sys.modules["manopth"].manolayer = sys.modules["manopth.manolayer"]
    for sub in ("consistdisplay", "rotateverts", "samplevis"):
        setattr(sys.modules["meshreg.visualize"], sub, sys.modules["meshreg.visualize." + sub])
    from meshreg.neurender import fastrender

    rng = np.random.default_rng(15)
    arrays, meta = {}, []
    for key, B, res, bg, crop in (("b1_bg", 1, (40, 28), 0.25, True), ("b2", 2, (36, 36), None, True),
                                  ("b2_nocrop", 2, (40, 24), None, False)):
        side = max(res)
        sc = scene(rng, B, side, hand_subdiv=1, obj_subdiv=0)
        verts = np.concatenate([sc["hand1"], sc["obj1"]], 1)
        faces = np.concatenate([sc["hand_faces"], sc["obj_faces"] + sc["hand1"].shape[1]], 0)[None].repeat(B, 0)
        colors = rng.uniform(0, 1, (B, verts.shape[1], 4)).astype(np.float32)  # (a fourth column: render takes [:, :, :3])
        out = fastrender.render(T(verts), T(faces), res, camintrs=T(sc["K1"]), colors=T(colors), bg_color=bg, crop_to_img=crop)
        arrays.update({f"{key}_verts": verts, f"{key}_faces": faces, f"{key}_K": sc["K1"], f"{key}_colors": colors,
                       f"{key}_out": N(out)})
        meta.append(dict(key=key, input_res=list(res), bg_color=bg, crop_to_img=crop, batch=B))
        print(key, tuple(out.shape), "covered", float((out[..., 3] > 0).float().mean()))
    raised = False
    try:
        fastrender.render(T(verts), T(faces), res, camintrs=T(sc["K1"]), colors=T(colors), bg_color=0.5)
    except RuntimeError:
        raised = True
    meta.append(dict(key="b2_bg_raises_in_the_reference", value=raised))
    print("B = 2 with bg_color raises in the reference:", raised)
    save("chain_fastrender.npz", arrays, meta)


if __name__ == "__main__":
    gens = dict(fastrender=gen_fastrender, rasterize=gen_rasterize, renderer=gen_renderer, opticalflow=gen_opticalflow,
                opticalflow_cfg=gen_opticalflow_config_sizes, warpbranch=gen_warpbranch)
    for name in (sys.argv[1:] or list(gens)):  # (every generator seeds its own rng: any subset reproduces its file)
        gens[name]()
