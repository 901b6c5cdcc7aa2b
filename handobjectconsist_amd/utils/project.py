"""Camera projection helper -- stands in for ``libyana.camutils.project``
(third-party, called at /root/reference/meshreg/warping/opticalflow.py:98-99)."""


def batch_proj2d(verts, camintr, camextr=None):
    """verts [B,V,3] (camera frame), camintr [B,3,3] -> pixel locations [B,V,2]."""
    if camextr is not None:
        verts = camextr[:, :3, :3].bmm(verts.transpose(1, 2)).transpose(1, 2) + camextr[:, :3, 3].unsqueeze(1)
    verts_hom2d = camintr.bmm(verts.transpose(1, 2)).transpose(1, 2)
    return verts_hom2d[:, :, :2] / verts_hom2d[:, :, 2:]
