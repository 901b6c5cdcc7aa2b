"""Golden vectors for the frame -> batch step, generated with the REAL libraries the reference's
host path runs through: Pillow's ``Image.transform(size, Image.AFFINE, coeffs)`` (NEAREST) on random frames
and on an all-white frame (the jitter mask, handobjset.py:361-362, 376-378), and torch's CPU
``float().div(255)`` (= torchvision ``to_tensor``) for the tensorisation.  Run from the repo root:

    python tests/golden/make_golden_augment.py        # writes tests/golden/augment_pil.npz

Pillow version used: see the ``pillow_version`` entry of the file."""
import os

import numpy as np
import PIL
import torch
from PIL import Image

rng = np.random.default_rng(20260928)
out = {"pillow_version": np.array(PIL.__version__), "torch_version": np.array(torch.__version__)}
cases = []
for case in range(36):
    Hs, Ws = int(rng.integers(3, 48)), int(rng.integers(3, 48))
    W, H = int(rng.integers(1, 40)), int(rng.integers(1, 40))
    kind = ("scale", "scale_int", "mirror", "rot", "rot", "crop", "crop_rot", "double", "double_big")[case % 9]
    if kind == "scale":
        a = [rng.uniform(0.2, 3), 0, rng.uniform(-15, 15), 0, rng.uniform(0.2, 3), rng.uniform(-15, 15)]
    elif kind == "scale_int":
        a = [float(rng.integers(1, 3)), 0, float(rng.integers(-4, 4)), 0, float(rng.integers(1, 3)), float(rng.integers(-4, 4))]
    elif kind == "mirror":
        a = [-rng.uniform(0.5, 2), 0, Ws - rng.uniform(0, 3), 0, rng.uniform(0.5, 2), rng.uniform(-3, 3)]
    elif kind == "rot":
        th, s = rng.uniform(-1, 1), rng.uniform(0.3, 2.5)
        a = [s * np.cos(th), -s * np.sin(th), rng.uniform(-10, 30), s * np.sin(th), s * np.cos(th), rng.uniform(-10, 30)]
    elif kind in ("crop", "crop_rot"):
        # the way the dataset builds them: float32 crop affine, float32 inverse
        cx, cy, sc = rng.uniform(0, Ws), rng.uniform(0, Hs), rng.uniform(4, 60)
        rot = 0.0 if kind == "crop" else rng.uniform(-0.6, 0.6)
        c, s_ = np.cos(rot), np.sin(rot)
        rotm = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1.0]])
        rc = rotm.dot([cx, cy, 1.0])[:2]
        crop = np.array([[W / sc, 0, W * (0.5 - rc[0] / sc)], [0, H / sc, H * (0.5 - rc[1] / sc)], [0, 0, 1.0]])
        inv = np.linalg.inv(crop.dot(rotm).astype(np.float32))
        a = [inv[0, 0], inv[0, 1], inv[0, 2], inv[1, 0], inv[1, 1], inv[1, 2]]
    elif kind == "double":
        a = [rng.uniform(0.5, 2), rng.uniform(-0.5, 0.5), rng.uniform(-10, 10) + 40000, rng.uniform(-0.5, 0.5),
             rng.uniform(0.5, 2), rng.uniform(-10, 10)]
    else:
        a = [rng.uniform(900, 2000), rng.uniform(-0.5, 0.5), rng.uniform(-10, 10), rng.uniform(-0.5, 0.5),
             rng.uniform(0.5, 2), rng.uniform(-10, 10)]
    a = [float(v) for v in a]
    src = rng.integers(1, 256, (Hs, Ws, 3), dtype=np.uint8)
    img = np.asarray(Image.fromarray(src).transform((W, H), Image.AFFINE, tuple(a)))
    white = np.asarray(Image.new("RGB", (Ws, Hs), (255, 255, 255)).transform((W, H), Image.AFFINE, tuple(a)))
    tens = torch.from_numpy(img.copy()).permute(2, 0, 1).contiguous().float().div(255)  # to_tensor
    tens = tens.sub(torch.tensor([0.5, 0.5, 0.5])[:, None, None]).div(torch.tensor([1.0, 1.0, 1.0])[:, None, None])
    mask = torch.from_numpy(white.copy()).permute(2, 0, 1).contiguous().float().div(255)
    out[f"c{case}_src"], out[f"c{case}_coeffs"] = src, np.array(a, np.float64)
    out[f"c{case}_size"] = np.array([W, H])
    out[f"c{case}_img"], out[f"c{case}_white"] = img, white
    out[f"c{case}_image"], out[f"c{case}_jittermask"] = tens.numpy(), mask.numpy()
    cases.append(kind)
out["kinds"] = np.array(cases)
# to_tensor of every byte value (CPU true division, not a multiplication by 1/255)
out["u8_div255"] = torch.arange(256, dtype=torch.uint8).float().div(255).numpy()
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "augment_pil.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path), "bytes")
