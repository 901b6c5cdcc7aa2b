"""CPU oracle of manopth's ``ManoLayer.forward`` -- TEST INFRASTRUCTURE, not product code.

Only ``tests/`` may import this module (the checker of ``csrc/mano_lbs.hip``); nothing under
``handobjectconsist_amd/`` does.

PARITY UNPINNED: ``manopth`` (git HEAD, un-pinned, /root/reference/environment.yml:34) and the licence-gated MANO
model files are absent from /root/reference and from this image, and the reference has no tests.  This file
restates the published algorithm of ``manopth/manolayer.py`` + ``rodrigues_layer.py`` + ``tensutils.py`` as SURVEY
appendix B.10 records it, anchored on the reference's call sites:
  /root/reference/meshreg/models/manobranch.py:70-85   (ManoLayer(ncomps=15, use_pca=True, flat_hand_mean=False,
                                                        center_idx=9, side=...) -- the model branch)
  /root/reference/meshreg/models/manobranch.py:130-136 (verts, joints = mano_layer(pose, th_betas=shape); millimetres,
                                                        divided by 1000 by the caller)
  /root/reference/meshreg/models/warpreg.py:54-60      (use_pca=False, flat_hand_mean=True, center_idx=None: faces only)

Written in manopth's OWN structure -- one joint after the other down the kinematic tree, per-joint 4x4 products,
per-vertex blend of the 16 transforms -- in numpy, parametrised by dtype: float64 serves as the reference the fp32
kernels are compared with and as the function whose central differences check their gradients (the product's
PyTorch restatement, ``SynthManoLayer.forward_torch``, arranges the same contractions as a few dense GEMMs; it is
checked against this file too, tests/test_oracle_mano.py)."""
import numpy as np

PARENTS = [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14]   # manopth: th_kintree_table[0]
TIPS_RIGHT = [745, 317, 444, 556, 673]                             # manolayer.py (side == "right")
REORDER = [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]


def quat2mat(q):
    """rodrigues_layer.quat2mat: normalised quaternion [N,4] (w,x,y,z) -> [N,3,3]."""
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    return np.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
                     2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
                     2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], 1).reshape(-1, 3, 3)


def batch_rodrigues(axisang):
    """rodrigues_layer.batch_rodrigues: [N,3] -> [N,3,3]; angle = |axisang + 1e-8|."""
    angle = np.linalg.norm(axisang + 1e-8, axis=1, keepdims=True)
    normed = axisang / angle
    half = angle * 0.5
    return quat2mat(np.concatenate([np.cos(half), np.sin(half) * normed], 1))


def _with_zeros(rot, tr):
    """tensutils.th_with_zeros: [3,3] rotation + [3] translation -> [4,4]."""
    out = np.zeros((4, 4), rot.dtype)
    out[:3, :3], out[:3, 3], out[3, 3] = rot, tr, 1.0
    return out


def mano_forward(c, pose_coeffs, betas=None, trans=None, ncomps=15, use_pca=True, center_idx=9, tips=TIPS_RIGHT,
                 dtype=np.float64):
    """``ManoLayer.forward(th_pose_coeffs, th_betas, th_trans)`` -> (verts[B,778,3], joints[B,21,3]) in millimetres.

    ``c``: the layer's buffers as arrays -- th_v_template [1,778,3], th_shapedirs [778,3,10], th_posedirs [778,3,135],
    th_J_regressor [16,778], th_weights [778,16], th_comps [45,45] (rows = components), th_hands_mean [1,45]."""
    f = lambda a: np.asarray(a, dtype)
    pose_coeffs = f(pose_coeffs)
    B = pose_coeffs.shape[0]
    betas = np.zeros((B, 10), dtype) if betas is None else f(betas)
    v_template, shapedirs, posedirs = f(c["th_v_template"])[0], f(c["th_shapedirs"]), f(c["th_posedirs"])
    j_reg, weights = f(c["th_J_regressor"]), f(c["th_weights"])
    comps, hands_mean = f(c["th_comps"])[:ncomps], f(c["th_hands_mean"]).reshape(45)
    verts_out, joints_out = np.zeros((B, 778, 3), dtype), np.zeros((B, 21, 3), dtype)
    for b in range(B):
        hand = pose_coeffs[b, 3:3 + ncomps] @ comps if use_pca else pose_coeffs[b, 3:48]
        full_pose = np.concatenate([pose_coeffs[b, :3], hands_mean + hand])          # [48]
        rots = batch_rodrigues(full_pose.reshape(16, 3))                               # [16,3,3]
        pose_map = (rots[1:] - np.eye(3, dtype=dtype)).reshape(135)
        v_shaped = shapedirs @ betas[b] + v_template                                   # [778,3]
        joints = j_reg @ v_shaped                                                      # [16,3]
        v_posed = v_shaped + posedirs @ pose_map
        # the chain, joint after joint (parents precede their children in MANO's numbering)
        G = [None] * 16
        G[0] = _with_zeros(rots[0], joints[0])
        for k in range(1, 16):
            G[k] = G[PARENTS[k]] @ _with_zeros(rots[k], joints[k] - joints[PARENTS[k]])
        G = np.stack(G)                                                                # [16,4,4]
        # remove the rest pose: G'_k = G_k - [0 | G_k (J_k; 0)]
        Gp = G.copy()
        for k in range(16):
            Gp[k, :, 3] -= G[k] @ np.concatenate([joints[k], [0.0]]).astype(dtype)
        T = np.einsum("vk,kij->vij", weights, Gp)                                      # [778,4,4]
        v_h = np.concatenate([v_posed, np.ones((778, 1), dtype)], 1)
        verts = np.einsum("vij,vj->vi", T, v_h)[:, :3]
        jtr = np.concatenate([G[:, :3, 3], verts[tips]], 0)[REORDER]                   # [21,3]
        if trans is None or float(np.linalg.norm(np.asarray(trans, np.float64))) == 0.0:  # manolayer.py: norm of the WHOLE tensor
            if center_idx is not None:
                center = jtr[center_idx].copy()
                jtr, verts = jtr - center, verts - center
        else:
            tb = f(trans)[b]
            jtr, verts = jtr + tb, verts + tb
        verts_out[b], joints_out[b] = verts * 1000, jtr * 1000
    return verts_out, joints_out


def directional_derivative(c, pose, betas, w_verts, w_joints, d_pose, d_betas, eps=1e-6, **kw):
    """Central difference, in float64, of L = <verts, w_verts> + <joints, w_joints> along (d_pose, d_betas): what
    <grad_pose, d_pose> + <grad_betas, d_betas> of an implementation's backward has to equal."""
    def loss(p, b_):
        v, j = mano_forward(c, p, b_, dtype=np.float64, **kw)
        return float((v * w_verts).sum() + (j * w_joints).sum())

    pose, betas = np.asarray(pose, np.float64), np.asarray(betas, np.float64)
    d_pose, d_betas = np.asarray(d_pose, np.float64), np.asarray(d_betas, np.float64)
    return (loss(pose + eps * d_pose, betas + eps * d_betas) - loss(pose - eps * d_pose, betas - eps * d_betas)) / (2 * eps)
