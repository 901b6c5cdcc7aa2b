"""Frame -> batch step on the CPU: the oracle's restatement of Pillow's nearest affine
transform against golden vectors made with the real Pillow (tests/golden/make_golden_augment.py) and against
the live library; the host-side crop geometry; the dataset wrapper's sampling logic."""
import os

import numpy as np
import pytest
import torch

from oracle import augment_ref as A

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "augment_pil.npz"))
N_CASES = len(GOLD["kinds"])


@pytest.mark.parametrize("case", range(N_CASES))
def test_oracle_matches_pillow_golden(case):
    src, coeffs = GOLD[f"c{case}_src"], GOLD[f"c{case}_coeffs"]
    W, H = GOLD[f"c{case}_size"]
    img, inside = A.pil_affine_nearest(src, coeffs, (W, H))
    assert np.array_equal(img, GOLD[f"c{case}_img"]), GOLD["kinds"][case]
    assert np.array_equal(inside, GOLD[f"c{case}_white"][..., 0] == 255)
    assert set(np.unique(GOLD[f"c{case}_white"])) <= {0, 255}
    # tensorisation: to_tensor + normalize(0.5, 1) as torch computes them on the CPU
    image = (img.astype(np.float32) / np.float32(255.0) - np.float32(0.5)).transpose(2, 0, 1)
    assert np.array_equal(image, GOLD[f"c{case}_image"])
    assert np.array_equal(np.broadcast_to(inside.astype(np.float32)[None], (3, H, W)), GOLD[f"c{case}_jittermask"])


def test_golden_covers_all_three_pillow_regimes():
    regimes = set()
    for case in range(N_CASES):
        a = GOLD[f"c{case}_coeffs"]
        W, H = GOLD[f"c{case}_size"]
        fits = all(abs(x * a[0] + y * a[1] + a[2]) < 32768 and abs(x * a[3] + y * a[4] + a[5]) < 32768
                   for x, y in ((0, 0), (W, H), (0, H), (W, 0)))
        regimes.add("scale" if a[1] == 0 and a[3] == 0 else ("fixed" if fits else "double"))
    assert regimes == {"scale", "fixed", "double"}


def test_to_tensor_is_a_true_division():
    got = np.arange(256, dtype=np.float32) / np.float32(255.0)
    assert np.array_equal(got, GOLD["u8_div255"])
    assert not np.array_equal(np.arange(256, dtype=np.float32) * np.float32(1.0 / 255.0), GOLD["u8_div255"])


def test_oracle_matches_live_pillow_randomised():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(7)
    for case in range(400):
        Hs, Ws, W, H = (int(v) for v in rng.integers(1, 40, 4))
        src = rng.integers(1, 256, (Hs, Ws, 3), dtype=np.uint8)
        if case % 3 == 0:
            a = [rng.uniform(0.2, 3), 0, rng.uniform(-20, 20), 0, rng.uniform(0.2, 3), rng.uniform(-20, 20)]
        elif case % 3 == 1:
            th, s = rng.uniform(-3, 3), rng.uniform(0.3, 2.5)
            a = [s * np.cos(th), -s * np.sin(th), rng.uniform(-20, 40), s * np.sin(th), s * np.cos(th), rng.uniform(-20, 40)]
        else:
            at, _ = A.get_affine_transform(np.array([rng.uniform(0, Ws), rng.uniform(0, Hs)]), rng.uniform(3, 80), (W, H),
                                           rot=rng.uniform(-0.5, 0.5) * (case % 2))
            a = list(A.inverse_coeffs(at))
        a = [float(v) for v in a]
        ref = np.asarray(Image.fromarray(src).transform((W, H), Image.AFFINE, tuple(a)))
        got, inside = A.pil_affine_nearest(src, a, (W, H))
        assert np.array_equal(ref, got), (case, a)
        assert np.array_equal(inside, (got != 0).any(-1))  # sources are >= 1 everywhere


def test_host_crop_geometry_matches_oracle_and_closed_forms():
    from handobjectconsist_amd.datasets import handutils

    rng = np.random.default_rng(3)
    for _ in range(50):
        center, scale = rng.uniform(50, 500, 2), rng.uniform(60, 400)
        res, rot = (int(rng.integers(64, 300)), int(rng.integers(64, 300))), rng.uniform(-1, 1)
        a1, p1 = handutils.get_affine_transform(center, scale, res, rot)
        a2, p2 = A.get_affine_transform(center, scale, res, rot)
        assert a1.dtype == np.float32 and np.array_equal(a1, a2) and np.array_equal(p1, p2)
        # the crop centre lands in the middle of the output, with and without rotation
        mid = handutils.transform_coords(center[None], a1)[0]
        assert np.allclose(mid, (res[1] / 2, res[0] / 2), atol=2e-3)
        # `scale` source pixels span the output
        a0, _ = handutils.get_affine_transform(center, scale, res, 0)
        edge = handutils.transform_coords(np.array([center - scale / 2, center + scale / 2]), a0)
        assert np.allclose(edge[1] - edge[0], (res[1], res[0]), atol=2e-3)
        pts = rng.uniform(0, 600, (7, 2))
        back = handutils.transform_coords(handutils.transform_coords(pts, a1), a1, invert=True)
        assert np.allclose(back, pts, atol=1e-2)
        assert np.array_equal(handutils.pil_coeffs(a1), np.array(A.inverse_coeffs(a1), np.float64))
        # post_rot_trans * K projects the ROTATED 3-D point where affinetrans sends the original projection
        K = np.array([[600.0, 0, res[1] / 2], [0, 600.0, res[0] / 2], [0, 0, 1]])
        X = np.array([0.05, -0.02, 0.5])
        c, s = np.cos(rot), np.sin(rot)
        Xr = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]]).dot(X)
        uv = K.dot(X)
        uv_r = p1.dot(K).dot(Xr)
        assert np.allclose(handutils.transform_coords((uv[:2] / uv[2])[None], a1)[0], uv_r[:2] / uv_r[2], atol=5e-2)


def _dataset(**kw):
    from handobjectconsist_amd.datasets import handobjset, synthpose

    ds = synthpose.SynthPoseDataset(3, seed=1, sides=("right", "left"))
    return ds, handobjset.HandObjSet(ds, inp_res=(256, 256), **kw)


def test_sequence_sampling_shares_the_augmentation():
    ds, hs = _dataset(sample_nb=3, spacing=2, block_rot=False, max_rot=0.4)
    assert hs.sequence_offsets() == [2, -2]
    hs.sample_nb = 5
    # the reference's loop advances the distance on BOTH branches (handobjset.py:404-418): +s, -s, +3s, -3s, ...
    assert hs.sequence_offsets() == [2, -2, 6, -6]
    hs.sample_nb = 3
    torch.manual_seed(0)
    seq = hs[2]
    assert len(seq) == 3 and [s["dist2query"] for s in seq] == [0, 1, 1]
    # same crop for every frame of the sequence (photometric consistency needs it)
    for other in seq[1:]:
        assert np.array_equal(other["affinetrans"], seq[0]["affinetrans"])
        assert "space_augm" not in other and "color_augm" not in other
    # the draws are torch's: same seed, same crop; other seed, other crop
    torch.manual_seed(0)
    again = hs[2]
    assert np.array_equal(again[0]["affinetrans"], seq[0]["affinetrans"])
    torch.manual_seed(1)
    assert not np.array_equal(hs[2][0]["affinetrans"], seq[0]["affinetrans"])


def test_augmentation_draw_order_and_ranges():
    ds, hs = _dataset(block_rot=False, max_rot=0.5, scale_jittering=0.3, center_jittering=0.2)
    center, scale = np.array([300.0, 200.0], np.float32), 150.0
    torch.manual_seed(5)
    aug = hs.draw_space_augm(center, scale)
    torch.manual_seed(5)
    cj = torch.distributions.uniform.Uniform(low=-1, high=1).sample((2,)).numpy()
    sj = torch.distributions.normal.Normal(0, 1).sample().item() + 1
    rot = torch.distributions.uniform.Uniform(low=-0.5, high=0.5).sample().item()
    assert np.array_equal(aug["center"], center + (0.2 * scale * cj).astype(int))
    assert aug["scale"] == scale * np.clip(0.3 * sj, 0.7, 1.3) and aug["rot"] == rot
    hs.train = False
    assert hs.draw_space_augm(center, scale) == {"rot": 0, "scale": scale, "center": center}


def test_flip_and_annotation_transforms():
    ds, hs = _dataset(sides="right", block_rot=True, train=True)
    torch.manual_seed(0)
    left = hs.get_sample(1)   # a left hand: mirrored to a right hand
    right = hs.get_sample(0)
    assert left["flip"] is True and left["side"] == "right" and right["flip"] is False
    c3 = ds.get_joints3d(1)
    c3[:, 0] = -c3[:, 0]
    assert np.allclose(left["joints3d"], c3 - c3[9], atol=1e-6)          # centred on joint 9 after mirroring
    assert np.allclose(left["joints3d"][9], 0)
    hv = ds.get_hand_verts3d(1)
    hv[:, 0] = -hv[:, 0]
    assert np.allclose(left["handverts3d"], hv - c3[9], atol=1e-6)
    # block_rot: no rotation whatever was drawn
    assert left["space_augm"]["rot"] == 0
    # rotation on: 3-D annotations rotate about the optical axis, intrinsics take the rotation-free crop
    hs2 = _dataset(sides="both", block_rot=False, max_rot=0.7)[1]
    torch.manual_seed(3)
    s = hs2.get_sample(0)
    rot = s["space_augm"]["rot"]
    assert rot != 0
    R = np.array([[np.cos(rot), -np.sin(rot), 0], [np.sin(rot), np.cos(rot), 0], [0, 0, 1]], np.float32)
    j = R.dot(ds.get_joints3d(0).T).T
    assert np.allclose(s["joints3d"], j - j[9], atol=1e-6)
    from handobjectconsist_amd.datasets import handutils

    _, post = handutils.get_affine_transform(s["space_augm"]["center"], s["space_augm"]["scale"], (256, 256), rot=rot)
    assert np.allclose(s["camintr"], post.dot(ds.get_camintr(0)), atol=1e-4)


def test_collated_sequence_batch_format():
    from handobjectconsist_amd.utils import collate

    ds, hs = _dataset(sample_nb=2, spacing=1, block_rot=True, sides="right")
    torch.manual_seed(0)
    batch = collate.seq_extend_collate([hs[i] for i in (0, 2, 5)], ["objverts3d", "objfaces", "objcanverts"])
    assert len(batch) == 2
    for frame in batch:
        assert frame["frame"].shape == (3, 480, 640, 3) and frame["frame"].dtype == torch.uint8
        assert frame["affinetrans"].shape == (3, 3, 3) and frame["flip"].shape == (3,)
        assert frame["camintr"].shape == (3, 3, 3) and frame["objfaces"].shape == (3, 2000, 3)
        assert frame["handverts3d"].shape == (3, 778, 3) and frame["joints3d"].shape == (3, 21, 3)


def test_default_color_fn_blurs_jitters_and_shares_parameters():
    """datasets/coloraugm.py (handobjset.py:339-358): the first frame of a sequence draws the colour parameters, the others get
    the SAME ones; the blur radius is drawn per frame from torch's generator; the jitter's draws are Python's ``random``."""
    import random

    from PIL import Image, ImageFilter

    from handobjectconsist_amd.datasets import coloraugm

    ds, hs = _dataset(sample_nb=2, spacing=1, block_rot=True)
    assert hs.color_fn is not None, "train mode applies the reference's colour augmentation by default"
    frame = np.ascontiguousarray(ds.get_image(0))
    random.seed(3)
    out, params = hs.color_fn(frame, hs, None, 0.7)
    assert out.shape == frame.shape and out.dtype == np.uint8 and not np.array_equal(out, frame)
    assert set(params) == {"sat", "bright", "contrast", "hue"}
    assert 1 - hs.brightness <= params["bright"] <= 1 + hs.brightness and abs(params["hue"]) <= hs.hue
    # handed-over parameters are used as they are (second frame of a sequence): no draw, same parameters back
    state = random.getstate()
    out2, params2 = hs.color_fn(frame, hs, params, 0.7)
    assert params2 == params
    # (the only draw left is the shuffle of the four operations)
    random.setstate(state)
    random.shuffle([0, 1, 2, 3])
    after_shuffle = random.getstate()
    random.setstate(state)
    hs.color_fn(frame, hs, params, 0.7)
    assert random.getstate() == after_shuffle
    # jitter off = the fixtures' generator stub: PIL's blur alone, neutral parameters
    plain, neutral = coloraugm.make_color_fn(jitter=False)(frame, hs, None, 0.7)
    assert np.array_equal(plain, np.asarray(Image.fromarray(frame).filter(ImageFilter.GaussianBlur(0.7))))
    assert neutral == {"sat": 1.0, "bright": 1.0, "contrast": 1.0, "hue": 0.0}
    # neutral factors leave the image alone up to the HSV round trip of the hue step; None switches a step off
    img = Image.fromarray(frame)
    assert np.array_equal(np.asarray(coloraugm.apply_jitter(img, brightness=1.0, contrast=1.0, saturation=1.0)), frame)
    assert np.array_equal(np.asarray(coloraugm.apply_jitter(img)), frame)
    # the hue shift is cyclic: +0.5 and -0.5 of a turn meet (int(127.5) = 127 steps either way: 254 = -2 mod 256)
    up, down = np.asarray(coloraugm.adjust_hue(img, 0.5), np.int32), np.asarray(coloraugm.adjust_hue(img, -0.5), np.int32)
    assert np.abs(up - down).max() <= 16  # (two hue steps of 256 apart, on saturated pixels)
    with pytest.raises(ValueError):
        coloraugm.adjust_hue(img, 0.6)
    # a whole sequence through the dataset: torch's stream is the only one that places the crop
    torch.manual_seed(0)
    a = hs[2]
    random.seed(11)
    torch.manual_seed(0)
    b = hs[2]
    assert np.array_equal(a[0]["affinetrans"], b[0]["affinetrans"]) and np.array_equal(a[1]["affinetrans"], b[1]["affinetrans"])
