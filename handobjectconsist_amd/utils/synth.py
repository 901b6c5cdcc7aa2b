"""Synthetic, seeded stand-ins for the data the hot path consumes.

The real inputs are licence-gated or absent (MANO pickles, FPHAB / HO3D frames and object
meshes: /root/reference/README.md:38-54), so tests, smoke() and bench.py use closed
triangle meshes with the reference's vertex / face counts: a 778-vertex /
1552-face "hand" (= MANO's 1538 faces + the 14 wrist-closing faces of
meshreg/models/manoutils.py:10-28; for any closed genus-0 mesh F = 2V - 4) and a
1002-vertex / 2000-face "object" (HO3D ``textured_simple_2000.obj`` size), random poses at
z in [0.35, 0.6] m, FPHAB-like intrinsics (f ~ 300-400 px, principal point 128 +- 8) and a
second frame = first frame + small motion (0-6 px flows).
"""
import functools

import numpy as np

HAND_VERTS, HAND_FACES = 778, 1552
OBJ_VERTS, OBJ_FACES = 1002, 2000
HAND_IGNORE_FACES = list(range(1538, 1552))  # manoutils.py:33


@functools.lru_cache(maxsize=None)
def sphere_mesh(n_verts):
    """Closed, outward-oriented triangulation of n_verts Fibonacci points on the unit sphere:
    verts [n,3] float32, faces [2n-4,3] int64."""
    from scipy.spatial import ConvexHull

    i = np.arange(n_verts, dtype=np.float64) + 0.5
    phi = np.arccos(1 - 2 * i / n_verts)
    theta = np.pi * (1 + 5 ** 0.5) * i
    v = np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], 1)
    hull = ConvexHull(v)
    f = hull.simplices.astype(np.int64)
    # orient outward: (v1 - v0) x (v2 - v0) . centroid > 0
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    flip = (np.cross(b - a, c - a) * (a + b + c)).sum(1) < 0
    f[flip] = f[flip][:, ::-1]
    # deterministic face order (ConvexHull's is arbitrary): sort by the smallest vertex id
    order = np.lexsort((f[:, 2], f[:, 1], f[:, 0]))
    f = f[order]
    assert f.shape[0] == 2 * n_verts - 4
    return v.astype(np.float32), f


# the 14 wrist-closing faces of meshreg/models/manoutils.py:10-28 and the 16 vertex ids they use
MANO_CLOSE_FACES = [
    [92, 38, 122], [234, 92, 122], [239, 234, 122], [279, 239, 122], [215, 279, 122], [215, 122, 118],
    [215, 118, 117], [215, 117, 119], [215, 119, 120], [215, 120, 108], [215, 108, 79], [215, 79, 78],
    [215, 78, 121], [214, 215, 121],
]
MANO_WRIST_IDS = sorted({v for f in MANO_CLOSE_FACES for v in f})


@functools.lru_cache(maxsize=None)
def hand_template():
    """778 verts / 1552 faces, palm-like flattened ellipsoid, metres, centred.

    Vertices are relabelled so that the 16 ids MANO's wrist-closing faces refer to are the 16
    vertices of the template's wrist-end cap: faces[:1538] play the role of MANO's open mesh and
    faces[1538:] ARE manoutils' 14 closing faces (a fan of a few centimetres, like the real ones)."""
    v, f = sphere_mesh(HAND_VERTS)
    v = v * np.array([0.045, 0.09, 0.018], np.float32)
    # a few bumps so that the silhouette is not convex ("fingers")
    v = v * (1.0 + 0.25 * np.cos(5 * np.arctan2(v[:, 0], v[:, 1] + 1e-6)) * (v[:, 1] > 0))[:, None]
    cap = np.argsort(v[:, 1])[: len(MANO_WRIST_IDS)]          # wrist end = smallest y
    perm = np.arange(HAND_VERTS)                              # new id -> old id
    free_new = [i for i in range(HAND_VERTS) if i not in set(MANO_WRIST_IDS)]
    free_old = [i for i in range(HAND_VERTS) if i not in set(cap.tolist())]
    perm[MANO_WRIST_IDS] = cap
    perm[free_new] = free_old
    inv = np.empty(HAND_VERTS, np.int64)
    inv[perm] = np.arange(HAND_VERTS)
    v, f = v[perm], inv[f]
    # drop the 14 faces closest to the wrist end, append the closing fan
    order = np.argsort(v[f].mean(1)[:, 1])
    keep = np.sort(order[14:])
    f = np.concatenate([f[keep], np.asarray(MANO_CLOSE_FACES, np.int64)], 0)
    assert f.shape == (HAND_FACES, 3)
    return v.astype(np.float32), f


@functools.lru_cache(maxsize=None)
def object_template():
    """1002 verts / 2000 faces, box-ish ellipsoid, metres, centred."""
    v, f = sphere_mesh(OBJ_VERTS)
    v = np.sign(v) * np.abs(v) ** 0.6 * np.array([0.035, 0.06, 0.035], np.float32)
    return v.astype(np.float32), f


def _rodrigues(rvec):
    """[B,3] axis-angle -> [B,3,3]."""
    th = np.linalg.norm(rvec, axis=1, keepdims=True) + 1e-12
    k = rvec / th
    K = np.zeros((rvec.shape[0], 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -k[:, 2], k[:, 1]
    K[:, 1, 0], K[:, 1, 2] = k[:, 2], -k[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -k[:, 1], k[:, 0]
    s, c = np.sin(th)[:, :, None], np.cos(th)[:, :, None]
    return np.eye(3)[None] + s * K + (1 - c) * (K @ K)


def random_scene(batch_size, seed=0, image_size=256, motion=True):
    """Two frames of a hand + object scene in camera coordinates.

    Returns dict of numpy arrays: verts1/verts2 [B,1780,3] (hand then object), faces
    [B,3552,3] int64, K1/K2 [B,3,3], plus the per-mesh pieces (hand_verts*, obj_verts*)."""
    rng = np.random.default_rng(seed)
    hv, hf = hand_template()
    ov, of = object_template()
    B = batch_size

    def place(template, rot, trans):
        return (template[None] @ np.transpose(rot, (0, 2, 1)) + trans[:, None]).astype(np.float32)

    scale = image_size / 256.0
    f = rng.uniform(300, 400, (B,)) * scale
    pp = (128 + rng.uniform(-8, 8, (B, 2))) * scale
    K = np.zeros((B, 3, 3), np.float32)
    K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2], K[:, 2, 2] = f, f, pp[:, 0], pp[:, 1], 1

    rot_h = rng.normal(0, 1.0, (B, 3))
    rot_o = rng.normal(0, 1.0, (B, 3))
    t_h = np.stack([rng.uniform(-0.03, 0.03, B), rng.uniform(-0.03, 0.03, B), rng.uniform(0.35, 0.6, B)], 1)
    t_o = t_h + rng.uniform(-0.05, 0.05, (B, 3))
    hand1 = place(hv, _rodrigues(rot_h), t_h)
    obj1 = place(ov, _rodrigues(rot_o), t_o)
    if motion:
        d = lambda s, n: rng.normal(0, s, (B, n))
        hand2 = place(hv, _rodrigues(rot_h + d(0.05, 3)), t_h + d(0.005, 3))
        obj2 = place(ov, _rodrigues(rot_o + d(0.05, 3)), t_o + d(0.005, 3))
    else:
        hand2, obj2 = hand1.copy(), obj1.copy()
    faces = np.concatenate([hf, of + HAND_VERTS], 0)[None].repeat(B, 0)
    return {
        "verts1": np.concatenate([hand1, obj1], 1), "verts2": np.concatenate([hand2, obj2], 1),
        "hand_verts1": hand1, "hand_verts2": hand2, "obj_verts1": obj1, "obj_verts2": obj2,
        "hand_faces": hf, "obj_faces": of, "faces": faces, "K1": K, "K2": K.copy(),
    }


def random_images(batch_size, height, width, seed=0):
    """image_ref, image in [-0.5, 0.5] (handobjset.py:372) and jitter masks = ones with a
    random 0-16 px zero border (handobjset.py:361-379), 3 identical channels."""
    rng = np.random.default_rng(seed + 12345)
    imgs = rng.uniform(-0.5, 0.5, (2, batch_size, 3, height, width)).astype(np.float32)
    # low-pass a little so that bilinear gradients are not pure noise
    imgs = (imgs + np.roll(imgs, 1, -1) + np.roll(imgs, 1, -2) + np.roll(imgs, (1, 1), (-1, -2))) / 4
    jm = np.ones((2, batch_size, 3, height, width), np.float32)
    for k in range(2):
        for b in range(batch_size):
            l, r, u, d = rng.integers(0, 17, size=4)
            if l: jm[k, b, :, :, :l] = 0
            if r: jm[k, b, :, :, -r:] = 0
            if u: jm[k, b, :, :u, :] = 0
            if d: jm[k, b, :, -d:, :] = 0
    return imgs[0], imgs[1], jm[0], jm[1]
