"""Host-side colour augmentation of a training frame: Gaussian blur + colour jitter (handobjset.py:339-358).

* The blur is PIL's ``ImageFilter.GaussianBlur`` on the (already mirrored) frame, exactly the reference's call: **pinned** --
  ``tests/golden/chain_dataset.npz`` holds a configuration with ``blur_radius > 0`` produced by the reference's
  ``HandObjSet`` with the real Pillow, and the mirror reproduces its images bit for bit.
* The jitter is ``libyana.transformutils.colortrans`` (``get_color_params`` + ``apply_jitter``; libyana@v0.2.0 is absent from
  the image and from /root/reference): **UNPINNED**, restated from the published algorithm -- factors drawn with Python's
  ``random`` (not torch's generator: a sample's position in torch's RNG stream does not depend on it), brightness /
  saturation / contrast through ``PIL.ImageEnhance``, hue as a cyclic shift of the H channel, applied in a shuffled order.
  The generator script of the fixtures stubs it with neutral parameters; ``make_color_fn(jitter=False)`` is that stub's twin."""
import random

import numpy as np
from PIL import Image, ImageEnhance, ImageFilter


def get_color_params(brightness=0, contrast=0, saturation=0, hue=0):
    """(brightness, contrast, saturation, hue) factors; ``None`` for a component that is switched off."""
    b = random.uniform(max(0, 1 - brightness), 1 + brightness) if brightness > 0 else None
    c = random.uniform(max(0, 1 - contrast), 1 + contrast) if contrast > 0 else None
    s = random.uniform(max(0, 1 - saturation), 1 + saturation) if saturation > 0 else None
    h = random.uniform(-hue, hue) if hue > 0 else None
    return b, c, s, h


def adjust_hue(img, hue_factor):
    """Cyclic shift of the hue channel by ``hue_factor`` of a turn (|hue_factor| <= 0.5)."""
    if not -0.5 <= hue_factor <= 0.5:
        raise ValueError("hue_factor is not in [-0.5, 0.5]")
    if img.mode in ("L", "1", "I", "F"):
        return img
    h, s, v = img.convert("HSV").split()
    np_h = ((np.array(h, dtype=np.int32) + int(hue_factor * 255)) & 255).astype(np.uint8)  # wraps: hue is cyclic
    return Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert(img.mode)


def apply_jitter(img, brightness=None, contrast=None, saturation=None, hue=None):
    ops = []
    if brightness is not None:
        ops.append(lambda im: ImageEnhance.Brightness(im).enhance(brightness))
    if saturation is not None:
        ops.append(lambda im: ImageEnhance.Color(im).enhance(saturation))
    if hue is not None:
        ops.append(lambda im: adjust_hue(im, hue))
    if contrast is not None:
        ops.append(lambda im: ImageEnhance.Contrast(im).enhance(contrast))
    random.shuffle(ops)
    for op in ops:
        img = op(img)
    return img


def make_color_fn(jitter=True):
    """``color_fn`` of ``HandObjSet``: (frame uint8 HWC as the reference sees it -- mirrored if the sample is --, dataset,
    colour parameters of the sequence's first frame or None, blur radius) -> (frame, colour parameters).  ``jitter=False``: the
    blur only, with the neutral parameters the fixtures' generator script puts in libyana's place."""

    def color_fn(frame, dataset, color_augm, blur_radius):
        img = Image.fromarray(np.ascontiguousarray(frame)).filter(ImageFilter.GaussianBlur(blur_radius))
        if color_augm is None:
            if jitter:
                bright, contrast, sat, hue = get_color_params(brightness=dataset.brightness, saturation=dataset.saturation,
                                                              hue=dataset.hue, contrast=dataset.contrast)
            else:
                bright, contrast, sat, hue = 1.0, 1.0, 1.0, 0.0
        else:
            sat, contrast, hue, bright = color_augm["sat"], color_augm["contrast"], color_augm["hue"], color_augm["bright"]
        if jitter:
            img = apply_jitter(img, brightness=bright, saturation=sat, hue=hue, contrast=contrast)
        # (np.array: a WRITABLE copy -- np.asarray of a PIL image is read-only, and unflipped samples hand that very array to
        # collate / torch.from_numpy)
        return np.array(img), {"sat": sat, "bright": bright, "contrast": contrast, "hue": hue}

    return color_fn
