"""A small ``pose_dataset`` for the dataset-glue fixtures (tests/golden/chain_dataset.npz): the accessor protocol of the
reference's dataset classes (meshreg/datasets/ho3dv2.py: get_image :299, get_joints3d :314, get_dist_idx :283, ...) over
the synthetic hand + object scenes of the package -- INPUTS only.  Objects differ in size from frame to frame (every
second scene keeps a part of its mesh), so that the collate's cyclic padding has something to do; sides alternate, so that
``sides="right"`` has left hands to mirror.  ``pil=True``: ``get_image`` returns a PIL image, as the reference's datasets do
(the generator script); ``pil=False``: the decoded uint8 array, as the package's GPU frame pipeline takes it."""
import numpy as np

from handobjectconsist_amd.datasets.synthpose import SynthPoseDataset


class FakePoseDataset(SynthPoseDataset):
    def __init__(self, num_pairs=3, seed=4, pil=False):
        super().__init__(num_pairs, frame_size=(160, 120), seed=seed, sides=("right", "left", "left", "right"))
        self.pil = pil
        # bring the principal points onto the small sensor (SynthPoseDataset laid them out for its default 640 x 480)
        for K in self.K:
            K[0, 2] += (160 - 640) / 2
            K[1, 2] += (120 - 480) / 2
        self.keep = [320 if (i // 2) % 2 == 0 else 200 for i in range(len(self.frames))]  # object vertices kept per frame

    def get_image(self, idx):
        if not self.pil:
            return self.frames[idx]
        from PIL import Image

        return Image.fromarray(self.frames[idx])

    def _faces(self, idx):
        f = self.obj_faces
        return f[(f < self.keep[idx]).all(1)]

    def get_obj_verts_trans(self, idx):
        return self.obj[idx][: self.keep[idx]].copy()

    def get_obj_faces(self, idx):
        return self._faces(idx)

    def get_obj_verts_can(self, idx):
        v, t, s = super().get_obj_verts_can(idx)
        return v[: self.keep[idx]], t, s


# (config name, HandObjSet keyword arguments, torch seed, indices asked for one after the other on that RNG stream)
CONFIGS = [
    ("train_single", dict(center_idx=9, sides="both", block_rot=False, max_rot=np.pi, sample_nb=None), 11, [0, 3, 4]),
    ("train_pair_right", dict(center_idx=9, sides="right", block_rot=False, max_rot=0.6, sample_nb=2, spacing=1), 12, [1, 2, 5]),
    ("train_triple_left_blockrot", dict(center_idx=9, sides="left", block_rot=True, sample_nb=3, spacing=2), 13, [0, 3]),
    ("val_single_nocenter", dict(center_idx=None, sides="both", train=False, sample_nb=None), 14, [2, 5]),
    ("train_mid_center", dict(center_idx=-1, sides="right", block_rot=False, max_rot=1.0, sample_nb=2, spacing=0,
                              scale_jittering=0.1, center_jittering=0.4), 15, [4, 1]),
    # Gaussian blur (handobjset.py:340-341: PIL's filter on the mirrored frame, a radius drawn per frame), both hand sides
    ("train_pair_blur", dict(center_idx=9, sides="right", block_rot=False, max_rot=0.3, sample_nb=2, spacing=1, blur_radius=2.0),
     16, [0, 1, 3]),
]
INP_RES = (64, 64)
