"""SURVEY 8 f4 pinned by RUNNING the reference: tests/golden/chain_dataset.npz holds what /root/reference's
``HandObjSet.get_sample`` / ``__getitem__`` (meshreg/datasets/handobjset.py:93-430) and ``seq_extend_collate``
(meshreg/datasets/collate.py:15-36, 77-83) returned around tests/dataset_fake.FakePoseDataset on seeded RNG streams
(generator: tests/golden/make_golden_dataset.py; only libyana / torchvision / torch._six are stubbed there).  The package's
mirror -- datasets/handobjset.HandObjSet + utils/collate -- must reproduce it: the augmentation draws and how far every
sample advances the RNG, the crop affine, ``post_rot_trans . K``, mirrored / rotated / centred 3-D annotations, the
sequence companions and their distances, the collated batch (cyclic padding, dtypes), and -- through the CPU oracle of the
frame -> tensor step here, through the HIP kernel in tests/test_gpu_frames.py -- the image and the jitter mask."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import augment_ref
from tests import dataset_fake

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chain_dataset.npz")
QUERIES = ("frame", "camintr", "joints3d", "handverts3d", "objverts3d", "objfaces", "objcanverts", "side")


def decode_image(level, offset):
    """The generator's lossless byte encoding of the reference's float images (make_golden_dataset.py)."""
    return level.astype(np.float32) / np.float32(255.0) - np.float32(offset)


def build(kw):
    from handobjectconsist_amd.datasets import coloraugm, handobjset

    ds = dataset_fake.FakePoseDataset(pil=False)
    return ds, handobjset.HandObjSet(ds, inp_res=dataset_fake.INP_RES, queries=QUERIES,
                                   color_fn=coloraugm.make_color_fn(jitter=False), **{"train": True, "blur_radius": 0.0, **kw})


@pytest.fixture(scope="module")
def golden():
    g = np.load(GOLDEN)
    return g, json.loads(str(g["meta"]))


@pytest.mark.parametrize("cname,kw,seed,idxs", dataset_fake.CONFIGS, ids=[c[0] for c in dataset_fake.CONFIGS])
def test_samples_match_the_reference_run(golden, cname, kw, seed, idxs):
    g, meta = golden
    info = meta["configs"][cname]
    ds, hs = build(kw)
    torch.manual_seed(seed)
    first = hs.get_sample(idxs[0])
    assert np.array_equal(np.asarray(first["space_augm"]["center"], np.float64), g[f"{cname}/space_center"])
    assert np.array_equal(np.asarray([first["space_augm"]["scale"], first["space_augm"]["rot"]], np.float64),
                          g[f"{cname}/space_scale_rot"])
    torch.manual_seed(seed)
    items = [hs[i] for i in idxs]
    # every sample advanced torch's RNG exactly as far as the reference's did
    assert np.array_equal(torch.rand(4).numpy(), g[f"{cname}/rng_after"]), "RNG stream position after the items"
    sides = []
    for n, item in enumerate(items):
        frames = item if isinstance(item, list) else [item]
        assert len(frames) == info["frames_per_item"]
        for k, sample in enumerate(frames):
            ref = lambda name: g[f"{cname}/item{n}/frame{k}/{name}"]  # noqa: E731
            assert "space_augm" not in sample and "color_augm" not in sample
            sides.append(sample["side"])
            assert int(sample["dist2query"]) == int(ref("dist2query"))
            assert np.array_equal(sample["affinetrans"], ref("affinetrans")), "crop affine"
            assert sample["camintr"].dtype == np.float32 and np.array_equal(sample["camintr"], ref("camintr"))
            for name in ("joints3d", "handverts3d", "objverts3d", "objcanverts"):
                assert sample[name].dtype == ref(name).dtype == np.float32, name
                assert np.array_equal(sample[name], ref(name)), name
            assert np.array_equal(sample["objfaces"], ref("objfaces"))
            assert np.array_equal(np.asarray(sample["objcantrans"]), ref("objcantrans")) and sample["objcanscale"] == float(ref("objcanscale"))
            if kw.get("center_idx", 9) is not None:
                assert np.array_equal(sample["center3d"], ref("center3d"))
            # the frame -> tensor step: the oracle of the GPU kernel on this sample's decoded frame
            image, mask = augment_ref.frame_to_tensors(sample["frame"], sample["affinetrans"], dataset_fake.INP_RES,
                                                       flip=sample["flip"])
            assert np.array_equal(image, decode_image(ref("image"), 0.5)), "image"
            assert np.array_equal(mask, decode_image(ref("jittermask"), 0.0)), "jitter mask"
    assert sides == info["sides"]


@pytest.mark.parametrize("cname,kw,seed,idxs", dataset_fake.CONFIGS, ids=[c[0] for c in dataset_fake.CONFIGS])
def test_collated_batch_matches_the_reference_run(golden, cname, kw, seed, idxs):
    from handobjectconsist_amd.utils import collate

    g, meta = golden
    info = meta["configs"][cname]
    ds, hs = build(kw)
    torch.manual_seed(seed)
    items = [hs[i] for i in idxs]
    ext = ["objverts3d", "objfaces", "objcanverts"]
    batch = collate.seq_extend_collate(items, ext) if isinstance(items[0], list) else [collate.extend_collate(items, ext)]
    assert len(batch) == info["frames_per_item"]
    for k, frame in enumerate(batch):
        assert list(frame["side"]) == info[f"collated_sides_frame{k}"]
        for name in ("affinetrans", "camintr", "joints3d", "handverts3d", "objverts3d", "objfaces", "objcanverts", "objcanscale",
                     "objcantrans", "dist2query"):
            ref = g[f"{cname}/collated/frame{k}/{name}"]
            got = frame[name]
            assert torch.is_tensor(got), name
            assert str(got.dtype) == info["collated_dtypes"][name], (name, got.dtype)
            assert tuple(got.shape) == ref.shape and np.array_equal(got.numpy(), ref), name
        assert frame["frame"].dtype == torch.uint8 and frame["frame"].shape[0] == len(idxs)
