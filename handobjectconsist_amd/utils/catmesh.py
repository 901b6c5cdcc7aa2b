"""Mesh concatenation -- stands in for ``libyana.renderutils.catmesh``
(third-party, called at /root/reference/meshreg/models/warpbranch.py:50)."""
import torch


def batch_cat_meshes(verts, faces, colors=None):
    """Concatenate meshes along the vertex / face dimension, offsetting face indices.
    verts: list of [B,Vi,3]; faces: list of [B,Fi,3] -> (verts [B,sum V,3], faces [B,sum F,3], colors)."""
    off, all_faces = 0, []
    for v, f in zip(verts, faces):
        all_faces.append(f + off)
        off += v.shape[1]
    all_colors = torch.cat(colors, 1) if colors is not None else None
    return torch.cat(verts, 1), torch.cat(all_faces, 1), all_colors
