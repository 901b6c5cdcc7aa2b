"""A synthetic ``pose_dataset`` with the accessor protocol of the reference's dataset classes
(meshreg/datasets/ho3dv2.py: get_image :299, get_joints3d :314, get_dist_idx :283, ...): frame pairs of
the synthetic hand + object scene as 'consecutive video frames' of a camera with a larger sensor than the
network input, so that the crop / augmentation path has real work to do.  Stands in for FPHAB / HO3D,
which cannot be shipped."""
import numpy as np

from handobjectconsist_amd.utils import synth


class SynthPoseDataset:
    has_dist2strong = False

    def __init__(self, num_pairs=4, frame_size=(640, 480), seed=0, sides=("right",)):
        """2 * num_pairs frames: frame 2k and 2k + 1 are the two time steps of scene k."""
        self.frame_size = tuple(frame_size)  # (W, H)
        W, H = self.frame_size
        scene = synth.random_scene(num_pairs, seed=seed, image_size=256)
        rng = np.random.default_rng(seed)
        self.hand, self.obj, self.K = [], [], []
        for k in range(num_pairs):
            for f in ("1", "2"):
                self.hand.append(scene["hand_verts" + f][k])
                self.obj.append(scene["obj_verts" + f][k])
                K = scene["K" + f][k].copy()
                K[0, 2] += (W - 256) / 2  # principal point of the larger sensor
                K[1, 2] += (H - 256) / 2
                self.K.append(K)
        self.obj_faces = scene["obj_faces"]
        self.frames = rng.integers(0, 256, (2 * num_pairs, H, W, 3), dtype=np.uint8)
        self.sides = [sides[i % len(sides)] for i in range(2 * num_pairs)]
        obj_all = np.concatenate(self.obj)
        self.can_trans = obj_all.mean(0)
        self.can_scale = float(np.linalg.norm(obj_all - self.can_trans, axis=1).max())

    def __len__(self):
        return len(self.frames)

    def get_image(self, idx):
        return self.frames[idx]

    def get_sides(self, idx):
        return self.sides[idx]

    def get_camintr(self, idx):
        return self.K[idx]

    def _proj(self, idx, pts):
        h = self.K[idx].dot(pts.T).T
        return h[:, :2] / h[:, 2:]

    def get_center_scale(self, idx):
        """Square box around the projected hand + object, 1.5x loose (what the datasets' own boxes are)."""
        p = self._proj(idx, np.concatenate([self.hand[idx], self.obj[idx]]))
        lo, hi = p.min(0), p.max(0)
        return ((lo + hi) / 2).astype(np.float32), float(1.5 * (hi - lo).max())

    def get_joints3d(self, idx):
        return self.hand[idx][:21].copy()

    def get_hand_verts3d(self, idx):
        return self.hand[idx].copy()

    def get_obj_verts_trans(self, idx):
        return self.obj[idx].copy()

    def get_obj_faces(self, idx):
        return self.obj_faces

    def get_obj_verts_can(self, idx):
        return (self.obj[idx] - self.can_trans) / self.can_scale, self.can_trans, self.can_scale

    def get_dist_idx(self, idx, dist=1):
        """Closest annotated frame `dist` steps ahead (> 0) or behind (< 0) inside the same scene."""
        other = idx ^ 1 if dist != 0 else idx
        return other, abs(other - idx)
