"""Affine crop geometry of the spatial augmentation -- stands in for ``libyana.transformutils.handutils``
(third party, libyana@v0.2.0, not vendored; call sites meshreg/datasets/handobjset.py:162, 173, 194, 209,
221, 364, 376).  Host-side numpy: these are 3x3 matrices per sample.  The image resampling itself is
``frames.frames_to_batch`` (GPU)."""
import numpy as np


def get_affine_trans_no_rot(center, scale, res):
    """[3,3] map of the square crop of side ``scale`` centred on ``center`` onto an output of ``res``."""
    sx, sy = float(res[1]) / scale, float(res[0]) / scale
    return np.array([[sx, 0.0, res[1] * (0.5 - float(center[0]) / scale)],
                     [0.0, sy, res[0] * (0.5 - float(center[1]) / scale)],
                     [0.0, 0.0, 1.0]])


def get_affine_transform(center, scale, res, rot=0):
    """-> (affinetrans, post_rot_trans), float32 [3,3].

    ``affinetrans`` maps source pixels to crop pixels (rotation by ``rot`` about the pixel origin followed by
    the crop around the rotated centre); ``post_rot_trans`` is the rotation-free crop that multiplies the
    camera intrinsics when the rotation is applied to the 3-D annotations instead (handobjset.py:180-183):
    its centre is ``center`` rotated about the middle of the output frame."""
    c, s = np.cos(rot), np.sin(rot)
    rot_mat = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    centre_h = np.array([float(center[0]), float(center[1]), 1.0])
    about_origin = rot_mat.dot(centre_h)[:2]
    to_mid = np.eye(3)
    to_mid[:2, 2] = (-res[1] / 2, -res[0] / 2)
    from_mid = np.eye(3)
    from_mid[:2, 2] = (res[1] / 2, res[0] / 2)
    about_mid = from_mid.dot(rot_mat).dot(to_mid).dot(centre_h)[:2]
    affinetrans = get_affine_trans_no_rot(about_origin, scale, res).dot(rot_mat)
    return affinetrans.astype(np.float32), get_affine_trans_no_rot(about_mid, scale, res).astype(np.float32)


def transform_coords(pts, affine_trans, invert=False):
    """2-D points [N,2] through a [3,3] affine (its inverse with ``invert``)."""
    mat = np.linalg.inv(affine_trans) if invert else affine_trans
    hom = np.concatenate([pts, np.ones((pts.shape[0], 1))], 1)
    return mat.dot(hom.transpose()).transpose()[:, :2]


def pil_coeffs(affine_trans):
    """The six ``Image.transform(..., Image.AFFINE, data)`` coefficients libyana's ``transform_img`` hands to
    Pillow: the rows of the INVERSE affine (output pixel -> source pixel).  The inverse is taken in the
    matrix's own dtype (float32 for the output of ``get_affine_transform``), as the host path does."""
    inv = np.linalg.inv(affine_trans)
    return np.array([inv[0, 0], inv[0, 1], inv[0, 2], inv[1, 0], inv[1, 1], inv[1, 2]], dtype=np.float64)
