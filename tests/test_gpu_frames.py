"""mr_frames_to_batch (decoded frames -> image / jittermask batch) against the oracle and against
the golden vectors made with the real Pillow + torch CPU path; byte / index work: BIT-EXACT."""
import os

import numpy as np
import pytest
import torch

from oracle import augment_ref as A

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "augment_pil.npz"))


def _run(cuda, frames, coeffs, size, flip=None, **kw):
    from handobjectconsist_amd.datasets import frames as F

    img, mask = F.frames_to_batch(torch.from_numpy(np.ascontiguousarray(frames)).to(cuda), np.asarray(coeffs, np.float64),
                                  size, flip=flip, **kw)
    return img.cpu().numpy(), None if mask is None else mask.cpu().numpy()


@pytest.mark.parametrize("case", range(len(GOLD["kinds"])))
def test_matches_pillow_golden(cuda, case):
    W, H = (int(v) for v in GOLD[f"c{case}_size"])
    img, mask = _run(cuda, GOLD[f"c{case}_src"][None], GOLD[f"c{case}_coeffs"][None], (W, H))
    assert np.array_equal(img[0], GOLD[f"c{case}_image"]), GOLD["kinds"][case]
    assert np.array_equal(mask[0], GOLD[f"c{case}_jittermask"])


def _random_coeffs(rng, kind, Ws, Hs, W, H):
    if kind == 0:
        a = [rng.uniform(0.2, 3), 0, rng.uniform(-20, 20), 0, rng.uniform(0.2, 3), rng.uniform(-20, 20)]
        if rng.random() < 0.3:
            a = [float(rng.integers(1, 3)), 0, float(rng.integers(-5, 5)), 0, float(rng.integers(1, 3)), float(rng.integers(-5, 5))]
    elif kind == 1:
        th, s = rng.uniform(-3, 3), rng.uniform(0.3, 2.5)
        a = [s * np.cos(th), -s * np.sin(th), rng.uniform(-20, 40), s * np.sin(th), s * np.cos(th), rng.uniform(-20, 40)]
    elif kind == 2:
        at, _ = A.get_affine_transform(np.array([rng.uniform(0, Ws), rng.uniform(0, Hs)]), rng.uniform(5, 80), (W, H),
                                       rot=0 if rng.random() < 0.5 else rng.uniform(-0.5, 0.5))
        a = list(A.inverse_coeffs(at))
    else:
        a = [rng.uniform(0.5, 2), rng.uniform(-0.5, 0.5), rng.uniform(-20, 20), rng.uniform(-0.5, 0.5), rng.uniform(0.5, 2),
             rng.uniform(-20, 20)]
        if rng.random() < 0.5:
            a[2] += 40000 * (1 if rng.random() < 0.5 else -1)
        else:
            a[0] *= 2000
    return [float(v) for v in a]


def test_batches_of_mixed_regimes_match_oracle(cuda):
    """Every frame of a batch has its own coefficients (and regime) and its own flip flag; widths that are not
    multiples of the 4-pixel store, 1-pixel outputs, frames larger and smaller than the output."""
    rng = np.random.default_rng(11)
    for trial in range(40):
        N, Hs, Ws = int(rng.integers(1, 7)), int(rng.integers(1, 70)), int(rng.integers(1, 70))
        W, H = int(rng.integers(1, 80)), int(rng.integers(1, 60))
        frames = rng.integers(0, 256, (N, Hs, Ws, 3), dtype=np.uint8)
        coeffs = [_random_coeffs(rng, int(rng.integers(0, 4)), Ws, Hs, W, H) for _ in range(N)]
        flip = rng.random(N) < 0.4
        mc = 1 if trial % 3 == 0 else 3
        img, mask = _run(cuda, frames, coeffs, (W, H), flip=flip, mask_channels=mc)
        for n in range(N):
            src = frames[n][:, ::-1] if flip[n] else frames[n]
            ref_u8, inside = A.pil_affine_nearest(src, coeffs[n], (W, H))
            ref = (ref_u8.astype(np.float32) / np.float32(255.0) - np.float32(0.5)).transpose(2, 0, 1)
            assert np.array_equal(img[n], ref), (trial, n, coeffs[n])
            assert np.array_equal(mask[n], np.broadcast_to(inside.astype(np.float32)[None], (mc, H, W)))


def test_dataset_crop_matches_oracle_frame_to_tensors(cuda):
    """The dataset's own route: float32 crop affine -> float32 inverse -> Pillow coefficients, 640x480 frames to
    256x256 inputs, with normalisation constants other than (0.5, 1)."""
    from handobjectconsist_amd.datasets import frames as F
    from handobjectconsist_amd.datasets import handutils

    rng = np.random.default_rng(5)
    N, Hs, Ws, res = 5, 480, 640, (256, 256)
    frames = rng.integers(0, 256, (N, Hs, Ws, 3), dtype=np.uint8)
    affs = np.stack([handutils.get_affine_transform(rng.uniform((200, 150), (440, 330)), rng.uniform(150, 500), res,
                                                    rot=(0, 0.3, 0, -0.2, 0)[n])[0] for n in range(N)])
    flip = np.array([False, True, True, False, False])
    img, mask = F.frames_to_batch(torch.from_numpy(frames).to(cuda), affs, res, flip=flip)
    for n in range(N):
        ref_img, ref_mask = A.frame_to_tensors(frames[n], affs[n], res, flip=bool(flip[n]))
        assert np.array_equal(img[n].cpu().numpy(), ref_img) and np.array_equal(mask[n].cpu().numpy(), ref_mask)
    assert 0.2 < float(mask.mean()) <= 1.0
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    img2, _ = F.frames_to_batch(torch.from_numpy(frames).to(cuda), affs, res, flip=flip, mean=mean, std=std, jittermask=False)
    u8, _ = A.transform_img(frames[0], affs[0], res)
    ref = (u8.astype(np.float32) / np.float32(255.0) - np.array(mean, np.float32)) / np.array(std, np.float32)
    assert np.array_equal(img2[0].cpu().numpy(), ref.transpose(2, 0, 1))


def test_identity_and_properties_at_full_size(cuda):
    """Size-independent properties at the metric's batch size (3 * 64 frames of 256x256 out of 640x480):
    identity coefficients copy the top-left window; the mask is exactly the indicator of `image != fill`;
    a frame shifted by k pixels gives the shifted crop."""
    from handobjectconsist_amd.datasets import frames as F

    rng = np.random.default_rng(2)
    N, Hs, Ws, W, H = 192, 480, 640, 256, 256
    frames = torch.from_numpy(rng.integers(1, 256, (N, Hs, Ws, 3), dtype=np.uint8)).to(cuda)
    ident = np.tile(np.array([1.0, 0, 0, 0, 1.0, 0]), (N, 1))
    img, mask = F.frames_to_batch(frames, ident, (W, H))
    # u8 / 255 - 0.5 as the CPU computes it (torch's GPU division by a scalar multiplies by the reciprocal)
    lut = torch.from_numpy(np.arange(256, dtype=np.float32) / np.float32(255.0) - np.float32(0.5)).to(cuda)
    ref = lut[frames[:, :H, :W].long()].permute(0, 3, 1, 2)
    assert torch.equal(img, ref) and bool((mask == 1).all())
    shift = ident.copy()
    shift[:, 2], shift[:, 5] = 500.0, -7.0   # source column x + 500, source row y - 7: partly outside
    img2, mask2 = F.frames_to_batch(frames, shift, (W, H))
    assert torch.equal(img2[:, :, 7:, :140], lut[frames[:, :H - 7, 500:640].long()].permute(0, 3, 1, 2))
    assert bool((mask2[:, :, :7] == 0).all()) and bool((mask2[:, :, 7:, 140:] == 0).all())
    assert bool((mask2[:, :, 7:, :140] == 1).all())
    assert torch.equal(mask2[:, 0] == 1, (img2 != -0.5).any(1))


def test_edge_cases_and_errors(cuda):
    from handobjectconsist_amd.datasets import frames as F

    frames = torch.zeros((2, 5, 6, 3), dtype=torch.uint8, device=cuda)
    ident = np.tile(np.array([1.0, 0, 0, 0, 1.0, 0]), (2, 1))
    img, mask = F.frames_to_batch(frames[:0], ident[:0], (4, 4))            # empty batch
    assert img.shape == (0, 3, 4, 4) and mask.shape == (0, 3, 4, 4)
    bad = ident.copy()
    bad[1, 2] = np.nan                                                    # non-finite coefficients: empty frame
    img, mask = F.frames_to_batch(frames + 9, bad, (4, 4))
    assert bool((mask[1] == 0).all()) and bool((img[1] == -0.5).all()) and bool((mask[0] == 1).all())
    with pytest.raises(TypeError):
        F.frames_to_batch(frames.cpu(), ident, (4, 4))
    with pytest.raises(ValueError):
        F.frames_to_batch(frames.float(), ident, (4, 4))
    with pytest.raises(ValueError):
        F.frames_to_batch(frames, ident[:1], (4, 4))
    with pytest.raises(RuntimeError):
        F.frames_to_batch(frames, ident, (4, 4), mask_channels=2)


def test_dataset_to_training_step(cuda):
    """HandObjSet -> seq_extend_collate -> assemble_batch -> WarpRegNet consistency step: the batch format the
    dataset side produces is the one the render + warp path consumes."""
    from handobjectconsist_amd.datasets import handobjset, synthpose
    from handobjectconsist_amd.models.synthnet import SynthMeshRegNet
    from handobjectconsist_amd.models.warpreg import WarpRegNet
    from handobjectconsist_amd.utils import collate

    res = (64, 64)
    ds = synthpose.SynthPoseDataset(3, frame_size=(320, 240), seed=4)
    hs = handobjset.HandObjSet(ds, inp_res=res, sample_nb=2, spacing=1, block_rot=True, sides="right", center_idx=9)
    torch.manual_seed(0)
    batch = collate.seq_extend_collate([hs[i] for i in (0, 2, 4)], ["objverts3d", "objfaces", "objcanverts"])
    samples = handobjset.assemble_batch(batch, cuda, res)
    for s, b in zip(samples, batch):
        assert s["image"].shape == (3, 3, 64, 64) and s["jittermask"].shape == (3, 3, 64, 64)
        assert s["image"].is_cuda and s["camintr"].is_cuda and "frame" not in s
        for n in range(3):
            ref_img, ref_mask = A.frame_to_tensors(b["frame"][n].numpy(), b["affinetrans"][n].numpy(), res,
                                                   flip=bool(b["flip"][n]))
            assert np.array_equal(s["image"][n].cpu().numpy(), ref_img)
            assert np.array_equal(s["jittermask"][n].cpu().numpy(), ref_mask)
    torch.manual_seed(0)
    model = SynthMeshRegNet().to(cuda).eval()
    pre = WarpRegNet(res, model, lambda_consist=0.5, lambda_data=0.5, criterion="l1", gt_refs=True, use_backward=True,
                     mano_faces=model.mano_layer.th_faces, pair_outputs="loss").to(cuda)
    pre.step_count = 1000
    loss, losses, _, _ = pre({"data": samples, "supervision": "consist"})
    loss.backward()
    assert torch.isfinite(loss) and "warp_consist" in losses
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)


def test_assembled_batches_equal_the_reference_run(cuda):
    """tests/golden/chain_dataset.npz -- the REFERENCE's HandObjSet.__getitem__ + seq_extend_collate run on CPU with the
    real Pillow in its image path (tests/golden/make_golden_dataset.py) -- against the package's GPU-side pipeline:
    HandObjSet (decoded frame + affine + flip) -> seq_extend_collate -> assemble_batch (ONE mr_frames_to_batch launch
    for all frames of the step): image and jitter mask bit for bit, the annotations as collated by the reference."""
    import json
    import os

    from handobjectconsist_amd.datasets import coloraugm, handobjset
    from handobjectconsist_amd.utils import collate
    from tests import dataset_fake
    from tests.test_oracle_dataset import QUERIES, decode_image

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chain_dataset.npz"))
    meta = json.loads(str(g["meta"]))
    for cname, kw, seed, idxs in dataset_fake.CONFIGS:
        ds = dataset_fake.FakePoseDataset(pil=False)
        hs = handobjset.HandObjSet(ds, inp_res=dataset_fake.INP_RES, queries=QUERIES,
                                   color_fn=coloraugm.make_color_fn(jitter=False), **{"train": True, "blur_radius": 0.0, **kw})
        torch.manual_seed(seed)
        items = [hs[i] for i in idxs]
        ext = ["objverts3d", "objfaces", "objcanverts"]
        batch = collate.seq_extend_collate(items, ext) if isinstance(items[0], list) else collate.extend_collate(items, ext)
        out = handobjset.assemble_batch(batch, cuda, dataset_fake.INP_RES)
        frames = out if isinstance(out, list) else [out]
        assert len(frames) == meta["configs"][cname]["frames_per_item"]
        for k, frame in enumerate(frames):
            for n in range(len(idxs)):
                ref = lambda name: g[f"{cname}/item{n}/frame{k}/{name}"]  # noqa: E731
                assert np.array_equal(frame["image"][n].cpu().numpy(), decode_image(ref("image"), 0.5)), (cname, k, n, "image")
                assert np.array_equal(frame["jittermask"][n].cpu().numpy(), decode_image(ref("jittermask"), 0.0)), (cname, k, n)
            for name in ("camintr", "joints3d", "handverts3d", "objverts3d", "objfaces", "objcanverts"):
                assert frame[name].is_cuda
                assert np.array_equal(frame[name].cpu().numpy(), g[f"{cname}/collated/frame{k}/{name}"]), (cname, k, name)
