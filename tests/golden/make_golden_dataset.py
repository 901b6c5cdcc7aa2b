"""Generate tests/golden/chain_dataset.npz by RUNNING THE REFERENCE'S OWN dataset glue on CPU.

Build container only (needs /root/reference).  The committed .npz is data: seeded inputs' outputs.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_dataset.py

What runs unmodified from /root/reference:
    meshreg/datasets/handobjset.py   HandObjSet.get_sample (:93-384: draw order of the augmentation, flip handling,
                                     post_rot_trans . K, rotation / centring of the 3-D annotations, image + jitter-mask
                                     path) and __getitem__ (:396-430: sequence sampling with shared augmentation)
    meshreg/datasets/collate.py      extend_collate / seq_extend_collate (:15-36, 77-83: cyclic padding + default_collate)
    meshreg/datasets/{queries,datutils}.py
around tests/dataset_fake.FakePoseDataset, with torch's RNG seeded per configuration.

What is stubbed (source NOT under /root/reference, packages absent from the image):
    libyana.transformutils.handutils  get_affine_transform / transform_coords -> oracle/augment_ref.py (restated, PARITY
                                      UNPINNED); transform_img -> the REAL Pillow ``Image.transform(res, AFFINE, rows of
                                      the inverse)`` (Pillow is installed)
    libyana.transformutils.colortrans get_color_params -> fixed neutral parameters, apply_jitter -> identity (the colour
                                      jitter is libyana's: the package restates it in datasets/coloraugm.py, unpinned;
                                      ASSUMED not to draw from torch's RNG)
The Gaussian blur in front of the jitter (:340-341) is the reference's own call on the REAL Pillow: the configurations run
with ``blur_radius`` 0 except ``train_pair_blur`` (2.0), which pins the package's default ``color_fn`` blur (on the mirrored
frame, a radius per frame from torch's generator).
    torchvision.transforms.functional to_tensor (uint8 HWC -> float CHW / 255), normalize ((x - mean) / std)
    torch._six                        container_abcs, string_classes, int_classes
"""
import collections.abc
import json
import os
import sys
import types
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
from oracle import augment_ref  # noqa: E402
from tests import dataset_fake  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def _module(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _transform_img(img, affine_trans, res):
    from PIL import Image

    t = np.linalg.inv(affine_trans)
    return img.transform(tuple(res), Image.AFFINE, (t[0, 0], t[0, 1], t[0, 2], t[1, 0], t[1, 1], t[1, 2]))


def _to_tensor(pic):
    arr = np.array(pic, np.uint8, copy=True)
    return torch.from_numpy(arr).permute(2, 0, 1).contiguous().float().div(255)


def _normalize(tensor, mean, std):
    mean, std = torch.as_tensor(mean, dtype=tensor.dtype), torch.as_tensor(std, dtype=tensor.dtype)
    return (tensor - mean[:, None, None]) / std[:, None, None]


def install():
    sys.path.insert(0, "/root/reference")
    hu = _module("libyana.transformutils.handutils", get_affine_transform=augment_ref.get_affine_transform,
                 transform_coords=augment_ref.transform_coords, transform_img=_transform_img)
    ct = _module("libyana.transformutils.colortrans", get_color_params=lambda **kw: (1.0, 1.0, 1.0, 0.0),
                 apply_jitter=lambda img, **kw: img)
    tu = _module("libyana.transformutils", handutils=hu, colortrans=ct)
    _module("libyana", transformutils=tu)
    fn = _module("torchvision.transforms.functional", to_tensor=_to_tensor, normalize=_normalize)
    tr = _module("torchvision.transforms", functional=fn)
    _module("torchvision", transforms=tr)
    _module("torch._six", container_abcs=collections.abc, string_classes=(str, bytes), int_classes=int)
    from meshreg.datasets import collate, handobjset, queries

    return handobjset, collate, queries


def main():
    handobjset, collate, Q = install()
    B, T = Q.BaseQueries, Q.TransQueries
    queries = [T.IMAGE, T.JITTERMASK, T.AFFINETRANS, T.CAMINTR, B.CAMINTR, T.JOINTS3D, B.JOINTS3D, T.HANDVERTS3D, T.OBJVERTS3D,
               B.OBJFACES, B.OBJCANVERTS, B.SIDE, T.CENTER3D]
    names = {T.IMAGE: "image", T.JITTERMASK: "jittermask", T.AFFINETRANS: "affinetrans", T.CAMINTR: "camintr",
             B.CAMINTR: "base_camintr", T.JOINTS3D: "joints3d", B.JOINTS3D: "base_joints3d", T.HANDVERTS3D: "handverts3d",
             T.OBJVERTS3D: "objverts3d", B.OBJFACES: "objfaces", B.OBJCANVERTS: "objcanverts", B.OBJCANSCALE: "objcanscale",
             B.OBJCANTRANS: "objcantrans", B.SIDE: "side", T.CENTER3D: "center3d", "dist2query": "dist2query"}
    arrays, meta = {}, {"configs": {}, "inp_res": list(dataset_fake.INP_RES)}
    for cname, kw, seed, idxs in dataset_fake.CONFIGS:
        ds = dataset_fake.FakePoseDataset(pil=True)
        # (TransQueries.CENTER3D with center_idx=None reads an unassigned local in the reference, handobjset.py:333-334)
        cfg_queries = [q for q in queries if not (q is T.CENTER3D and kw.get("center_idx", 9) is None)]
        hs = handobjset.HandObjSet(ds, inp_res=dataset_fake.INP_RES, queries=cfg_queries,
                                   **{"train": True, "blur_radius": 0.0, **kw})
        # (1) get_sample alone: the augmentation it drew, for the first index
        torch.manual_seed(seed)
        first = hs.get_sample(idxs[0])
        sa = first["space_augm"]
        arrays[f"{cname}/space_center"] = np.asarray(sa["center"], np.float64)
        arrays[f"{cname}/space_scale_rot"] = np.asarray([sa["scale"], sa["rot"]], np.float64)
        # (2) __getitem__ for every index on ONE RNG stream (pins how many draws a sample consumes)
        torch.manual_seed(seed)
        items = [hs[i] for i in idxs]
        arrays[f"{cname}/rng_after"] = torch.rand(4).numpy()  # where the stream stands afterwards
        sides = []
        for n, item in enumerate(items):
            frames = item if isinstance(item, list) else [item]
            for k, sample in enumerate(frames):
                assert "space_augm" not in sample and "color_augm" not in sample
                for key, val in sample.items():
                    if key is B.SIDE:
                        sides.append(val)
                        continue
                    if val is None:
                        continue
                    arr = val.numpy() if torch.is_tensor(val) else np.asarray(val)
                    if key in (T.IMAGE, T.JITTERMASK):
                        # lossless byte encoding of the float image: stored as the u8 level, decoded by
                        # tests' `decode_image` = float32(level) / 255 - offset, checked bit for bit here
                        off = np.float32(0.5 if key is T.IMAGE else 0.0)
                        level = np.rint((arr.astype(np.float64) + float(off)) * 255.0).astype(np.uint8)
                        assert np.array_equal(level.astype(np.float32) / np.float32(255.0) - off, arr), "encoding is not lossless"
                        arr = level
                    arrays[f"{cname}/item{n}/frame{k}/{names[key]}"] = arr
        meta["configs"][cname] = {"idxs": idxs, "seed": seed, "frames_per_item": len(items[0]) if isinstance(items[0], list) else 1,
                                  "sides": sides}
        # (3) the collated batch of those items
        torch.manual_seed(seed)
        items = [hs[i] for i in idxs]
        ext = [T.OBJVERTS3D, B.OBJFACES, B.OBJCANVERTS]
        batch = collate.seq_extend_collate(items, ext) if isinstance(items[0], list) else [collate.extend_collate(items, ext)]
        for k, frame in enumerate(batch):
            for key, val in frame.items():
                if key is B.SIDE:
                    meta["configs"][cname][f"collated_sides_frame{k}"] = list(val)
                elif key not in (T.IMAGE, T.JITTERMASK):  # (the images are in the per-sample arrays already)
                    arrays[f"{cname}/collated/frame{k}/{names[key]}"] = val.numpy()
                    meta["configs"][cname].setdefault("collated_dtypes", {})[names[key]] = str(val.dtype)
    arrays["meta"] = np.array(json.dumps(meta))
    path = os.path.join(OUT, "chain_dataset.npz")
    np.savez_compressed(path, **arrays)
    print(f"chain_dataset.npz: {len(arrays)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
