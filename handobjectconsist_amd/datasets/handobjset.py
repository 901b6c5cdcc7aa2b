"""Hand-object frame sequences -> samples for the consistency trainer; counterpart of
meshreg/datasets/handobjset.py (same constructor arguments, same augmentation draws, same sequence
sampling) re-cut for a GPU-side image path:

* a sample carries the DECODED frame (uint8 HWC) together with its crop affine and flip flag instead of the
  transformed image; ``assemble_batch`` uploads the collated frames and produces ``image`` / ``jittermask``
  for the whole batch with one kernel (``frames.frames_to_batch``, bit-exact with the PIL path of
  handobjset.py:361-379);
* samples are dicts keyed by plain strings (the names of the reference's query enums in lower case:
  TransQueries.IMAGE -> "image", TransQueries.JITTERMASK -> "jittermask", TransQueries.CAMINTR -> "camintr",
  TransQueries.JOINTS3D -> "joints3d", TransQueries.HANDVERTS3D -> "handverts3d", TransQueries.OBJVERTS3D ->
  "objverts3d", BaseQueries.OBJFACES -> "objfaces", BaseQueries.OBJCANVERTS -> "objcanverts", ...).

``pose_dataset`` is any object with the accessor protocol of the reference's dataset classes
(ho3dv2.py / fhbhands.py): get_image, get_center_scale, get_camintr, get_joints3d, get_hand_verts3d,
get_obj_verts_trans, get_obj_faces, get_obj_verts_can, get_sides, get_dist_idx.  Colour jitter and blur
(libyana colortrans + PIL filters on the host, handobjset.py:339-358) are a host callable: ``datasets/coloraugm.py``
by default (blur pinned by a fixture, jitter restated), ``color_fn=None`` switches them off."""
import random
import traceback

import numpy as np
import torch
from torch.distributions.normal import Normal
from torch.distributions.uniform import Uniform
from torch.utils.data import Dataset

from handobjectconsist_amd.datasets import frames as frames_mod
from handobjectconsist_amd.datasets import handutils

DEFAULT_QUERIES = ("frame", "camintr", "joints3d", "handverts3d", "objverts3d", "objfaces", "objcanverts", "side")


def flip_hand_side(target_side, hand_side):
    """datutils.flip_hand_side (datutils.py:1-12): mirror left hands to right (or the opposite) on request."""
    if target_side in ("right", "left") and hand_side != target_side:
        return target_side, True
    return hand_side, False


class HandObjSet(Dataset):
    def __init__(self, pose_dataset, center_idx=9, inp_res=(256, 256), max_rot=np.pi, normalize_img=False,
                 split="train", scale_jittering=0.3, center_jittering=0.2, train=True, hue=0.15, saturation=0.5,
                 contrast=0.5, brightness=0.5, blur_radius=0.5, spacing=2, queries=DEFAULT_QUERIES, sides="both",
                 block_rot=False, sample_nb=None, has_dist2strong=False, color_fn="reference"):
        self.pose_dataset = pose_dataset
        self.center_idx, self.inp_res = center_idx, tuple(inp_res)
        self.normalize_img, self.sides = normalize_img, sides
        self.sample_nb, self.spacing = sample_nb, spacing
        self.hue, self.contrast, self.brightness, self.saturation = hue, contrast, brightness, saturation
        self.blur_radius = blur_radius
        self.max_rot, self.block_rot = max_rot, block_rot
        self.train, self.scale_jittering, self.center_jittering = train, scale_jittering, center_jittering
        self.queries = tuple(queries)
        self.has_dist2strong = has_dist2strong
        if color_fn == "reference":  # the reference's own augmentation (handobjset.py:339-358)
            from handobjectconsist_amd.datasets import coloraugm

            color_fn = coloraugm.make_color_fn(jitter=True)
        self.color_fn = color_fn  # (frame_u8, dataset, color_augm | None, blur_radius) -> (frame_u8, color_augm)

    def __len__(self):
        return len(self.pose_dataset)

    # ---- augmentation draws (handobjset.py:130-157): same distributions, same order of draws
    def draw_space_augm(self, center, scale):
        if not self.train:
            return {"rot": 0, "scale": scale, "center": center}
        center_jit = Uniform(low=-1, high=1).sample((2,)).numpy()
        center = center + (self.center_jittering * scale * center_jit).astype(int)
        scale_jit = Normal(0, 1).sample().item() + 1
        factor = np.clip(self.scale_jittering * scale_jit, 1 - self.scale_jittering, 1 + self.scale_jittering)
        rot = Uniform(low=-self.max_rot, high=self.max_rot).sample().item()
        return {"rot": rot, "scale": scale * factor, "center": center}

    def get_sample(self, idx, query=None, color_augm=None, space_augm=None):
        ds, q = self.pose_dataset, (self.queries if query is None else query)
        sample = {}
        hand_side, flip = flip_hand_side(self.sides, ds.get_sides(idx)) if "side" in q else (None, False)
        if hand_side is not None:
            sample["side"] = hand_side
        want_img = "frame" in q
        if want_img:
            center, scale = ds.get_center_scale(idx)
            frame = np.asarray(ds.get_image(idx))
            width = frame.shape[1]
            if flip:
                center = np.array(center).copy()
                center[0] = width - center[0]
            if space_augm is None:
                space_augm = self.draw_space_augm(center, scale)
        elif space_augm is None:
            space_augm = {"rot": 0, "scale": None, "center": None}
        rot = 0 if self.block_rot else space_augm["rot"]
        space_augm = dict(space_augm, rot=rot)
        sample["space_augm"] = space_augm
        rot_mat = np.array([[np.cos(rot), -np.sin(rot), 0], [np.sin(rot), np.cos(rot), 0], [0, 0, 1]]).astype(np.float32)
        if want_img:
            affinetrans, post_rot_trans = handutils.get_affine_transform(space_augm["center"], space_augm["scale"],
                                                                        self.inp_res, rot=rot)
            sample["affinetrans"] = affinetrans
            if self.train:
                # the blur radius is drawn for EVERY training frame, also for the companions of a sequence that inherit
                # their colour parameters (handobjset.py:341): part of how far a sample advances torch's RNG stream
                blur_radius = Uniform(low=0, high=1).sample().item() * self.blur_radius
                if self.color_fn is not None:
                    # (the reference blurs the MIRRORED image, handobjset.py:120-122 before :341; the frame travels unmirrored
                    # to the GPU kernel, which flips on the fly: mirror, augment, mirror back -- PIL's box-blur passes are
                    # not symmetric to the last bit)
                    view = frame[:, ::-1] if flip else frame
                    view, color_augm = self.color_fn(view, self, color_augm, blur_radius)
                    frame = view[:, ::-1] if flip else view
            sample["color_augm"] = color_augm if self.train else None
            sample["frame"] = np.ascontiguousarray(frame)
            sample["flip"] = bool(flip)
        if "camintr" in q:
            camintr = ds.get_camintr(idx)
            # the rotation is applied to the 3-D annotations: only the crop multiplies the intrinsics (:180-183)
            sample["camintr"] = (post_rot_trans.dot(camintr) if want_img else camintr).astype(np.float32)

        def mirrored(pts):
            pts = np.array(pts, dtype=np.float32)
            if flip:
                pts[:, 0] = -pts[:, 0]
            return pts

        def rotated(pts):
            return rot_mat.dot(pts.transpose(1, 0)).transpose()

        center3d = None
        if any(k in q for k in ("joints3d", "handverts3d", "objverts3d")):
            joints3d = mirrored(ds.get_joints3d(idx))
            if self.train:
                joints3d = rotated(joints3d)
            if self.center_idx is not None:
                center3d = (joints3d[9] + joints3d[0]) / 2 if self.center_idx == -1 else joints3d[self.center_idx]
            if "joints3d" in q:
                sample["joints3d"] = (joints3d - center3d if center3d is not None else joints3d).astype(np.float32)
            sample["center3d"] = None if center3d is None else center3d.astype(np.float32)
        for key, getter in (("handverts3d", "get_hand_verts3d"), ("objverts3d", "get_obj_verts_trans")):
            if key in q:
                pts = rotated(mirrored(getattr(ds, getter)(idx)))
                sample[key] = (pts - center3d if center3d is not None else pts).astype(np.float32)
        if "objfaces" in q:
            sample["objfaces"] = np.asarray(ds.get_obj_faces(idx))
        if "objcanverts" in q:
            canverts, cantrans, canscale = ds.get_obj_verts_can(idx)
            sample["objcanverts"] = mirrored(canverts)
            sample["objcanscale"], sample["objcantrans"] = canscale, cantrans
        return sample

    def get_safesample(self, idx, color_augm=None, space_augm=None):
        """A frame that fails to load is replaced by a neighbour within +-10 (handobjset.py:386-394)."""
        try:
            return self.get_sample(idx, color_augm=color_augm, space_augm=space_augm)
        except Exception:
            traceback.print_exc()
            other = random.randint(max(0, idx - 10), min(len(self), idx + 10))
            print(f"Encountered error processing sample {idx}, trying {other} instead")
            return self.get_sample(other)

    def sequence_offsets(self):
        """Signed frame distances of the sample_nb - 1 companions, as the reference's loop produces them
        (handobjset.py:404-418; the distance grows on both branches): +s, -s, +3s, -3s, +5s, ..."""
        offs, dist = [], 0
        for k in range((self.sample_nb or 1) - 1):
            if k % 2 == 0:
                dist += self.spacing
                offs.append(dist)
            else:
                offs.append(-dist)
                dist += self.spacing
        return offs

    def __getitem__(self, idx):
        sample = self.get_safesample(idx)
        sample["dist2query"] = 0
        space_augm, color_augm = sample.pop("space_augm"), sample.pop("color_augm", None)
        if self.sample_nb is None:
            return sample
        samples = [sample]
        for off in self.sequence_offsets():
            next_idx, dist2query = self.pose_dataset.get_dist_idx(idx, dist=off)
            # companions share the augmentation of the query frame so that photometric consistency holds
            other = self.get_safesample(next_idx, color_augm=color_augm, space_augm=space_augm)
            other["dist2query"] = dist2query
            other.pop("space_augm")
            other.pop("color_augm", None)
            samples.append(other)
        return samples


def assemble_batch(batch, device, inp_res, normalize_img=False, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """Collated batch (one frame's dict, or a list of them from ``seq_extend_collate``) -> device-resident
    tensors with ``image`` / ``jittermask`` built by the GPU from ``frame`` / ``affinetrans`` / ``flip``.
    All frames of the step go through ONE ``frames_to_batch`` launch."""
    dicts = batch if isinstance(batch, (list, tuple)) else [batch]
    out = []
    for d in dicts:
        out.append({k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in d.items()
                    if k not in ("frame", "affinetrans", "flip")})
    with_frames = [i for i, d in enumerate(dicts) if "frame" in d]
    if with_frames:
        frames = torch.cat([torch.as_tensor(dicts[i]["frame"]) for i in with_frames], 0).to(device, non_blocking=True)
        affines = np.concatenate([np.asarray(dicts[i]["affinetrans"]) for i in with_frames], 0)
        flips = np.concatenate([np.asarray(dicts[i]["flip"]).reshape(-1) for i in with_frames], 0)
        m, s = (mean, std) if normalize_img else ((0.5, 0.5, 0.5), (1.0, 1.0, 1.0))
        image, mask = frames_mod.frames_to_batch(frames, affines, inp_res, flip=flips, mean=m, std=s)
        lo = 0
        for i in with_frames:
            n = len(dicts[i]["frame"])
            out[i]["image"], out[i]["jittermask"] = image[lo:lo + n], mask[lo:lo + n]
            lo += n
    return out if isinstance(batch, (list, tuple)) else out[0]
