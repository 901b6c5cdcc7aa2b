"""Mesh-induced optical flow -- drop-in for meshreg/warping/opticalflow.py.

Same two functions and signatures as the reference (opticalflow.py:10, :51).  The flow is
the per-vertex 2-D displacement ``proj(v_t2) - proj(v_t1)`` painted as a vertex texture
``(dx, dy, 1)`` and rendered with the HIP rasteriser; masks follow the reference's algebra
including its quirks (Q2 un-flipped face_index_map flipped by hand, Q3
ignore list indexes the pre-fill-back faces, Q4 ``mask_flow2`` reset to raw alpha inside
the occlusion block).
"""
from typing import List

import torch

from handobjectconsist_amd.utils import project, textutils
from handobjectconsist_amd.warping import imgflowarp


def get_opticalflows(
    verts_cam: List[torch.Tensor],
    faces: torch.Tensor,
    camintrs: List[torch.Tensor],
    neurenderer,
    orig_img_size=None,
    detach_textures: bool = False,
    detach_renders: bool = False,
    ignore_face_idxs=None,
):
    """
    Compute optical flow between pairs of meshes (the same mesh at different time steps),
    always comparing to the first mesh (reference opticalflow.py:10-48).
    """
    all_flows = []
    for vert_world, camintr in zip(verts_cam[1:], camintrs[1:]):
        flows = get_opticalflow(
            [verts_cam[0], vert_world],
            faces,
            [camintrs[0], camintr],
            neurenderer,
            orig_img_size=orig_img_size,
            detach_textures=detach_textures,
            detach_renders=detach_renders,
            ignore_face_idxs=ignore_face_idxs,
        )
        all_flows.append(flows)
    return all_flows


# Use the renderer's fused vertex-colour path when it has one and the positions are detached
# (False: always build the face textures and call neurenderer(...) like the reference).
USE_VERTEX_COLOR_RENDER = True


def _render_flow(neurenderer, verts, faces, sample_flows, camintr, detach_textures, detach_renders):
    """neurenderer(verts, faces, batch_vertex_textures(faces, sample_flows), K=..., detach_renders=...)
    (opticalflow.py:103-108)."""
    if (USE_VERTEX_COLOR_RENDER and detach_renders and hasattr(neurenderer, "render_vertex_colors")
            and getattr(neurenderer, "no_light", False) and getattr(neurenderer, "camera_mode", "") == "projection"):
        cols = sample_flows.detach() if detach_textures else sample_flows
        return neurenderer.render_vertex_colors(verts, faces, cols, K=camintr)
    all_textures = textutils.batch_vertex_textures(faces, sample_flows)
    if detach_textures:
        all_textures = all_textures.detach()
    return neurenderer(verts, faces, all_textures, K=camintr, detach_renders=detach_renders)


def _ignore_mask(face_index_map, ignore_face_idxs):
    """1 where the winning face is not in the ignore list, in IMAGE orientation
    (opticalflow.py:110-116: |fim - ids|.min != 0, then the manual vertical flip).  Done as a
    table lookup over face indices instead of materialising the [B, is, is, 14] difference."""
    ids = torch.as_tensor(ignore_face_idxs, dtype=torch.long, device=face_index_map.device)
    n = int(max(int(ids.max().item()) + 2, 2)) if ids.numel() else 2
    key = (tuple(int(i) for i in ignore_face_idxs), str(face_index_map.device))
    lut = _LUT_CACHE.get(key)
    if lut is None:
        lut = torch.ones(n, dtype=torch.float32, device=face_index_map.device)
        lut[ids[ids >= 0] + 1] = 0.0  # slot 0 = background (-1)
        _LUT_CACHE[key] = lut
    idx = (face_index_map.long() + 1).clamp_(max=lut.numel() - 1)
    hi = face_index_map >= (lut.numel() - 1)  # faces beyond the table are never ignored
    keep = torch.where(hi, torch.ones((), device=lut.device), lut[idx])
    return keep.flip(1).unsqueeze(1)


_LUT_CACHE = {}


def get_opticalflow(
    verts_cam: List[torch.Tensor],
    faces: torch.Tensor,
    camintrs: List[torch.Tensor],
    neurenderer,
    orig_img_size=None,
    mask_occlusions: bool = True,
    detach_textures: bool = False,
    detach_renders: bool = True,
    ignore_face_idxs=None,
):
    """
    Compute optical flow in image space given the displacement of the vertices in
    verts_cam (reference opticalflow.py:51-156).

    Returns:
        [pred_flow12, pred_flow21], each [batch_size, H, W, 2] in pixel units.
    """
    gt_locs2d_1 = project.batch_proj2d(verts_cam[0], camintrs[0])
    gt_locs2d_2 = project.batch_proj2d(verts_cam[1], camintrs[1])
    # forward optical flow
    verts_displ2d_12 = gt_locs2d_2 - gt_locs2d_1
    sample_flows = torch.cat([verts_displ2d_12, torch.ones_like(verts_displ2d_12[:, :, :1])], -1)
    renderout = _render_flow(neurenderer, verts_cam[0], faces, sample_flows, camintrs[0], detach_textures,
                             detach_renders)
    mask_flow1 = (renderout["alpha"].unsqueeze(1) > 0.99999).float()
    if ignore_face_idxs is not None:
        mask_flow1 = mask_flow1 * _ignore_mask(renderout["face_index_map"], ignore_face_idxs)
    pred_flow12 = renderout["rgb"] * mask_flow1

    # backward optical flow
    verts_displ2d_21 = gt_locs2d_1 - gt_locs2d_2
    sample_flows = torch.cat([verts_displ2d_21, torch.ones_like(verts_displ2d_21[:, :, :1])], -1)
    # (the reference never detaches the second texture set, opticalflow.py:123)
    renderout = _render_flow(neurenderer, verts_cam[1], faces, sample_flows, camintrs[1], False, detach_renders)
    mask_flow2 = (renderout["alpha"].unsqueeze(1) > 0.99999).float()
    if ignore_face_idxs is not None:
        mask_flow2 = mask_flow2 * _ignore_mask(renderout["face_index_map"], ignore_face_idxs)
    pred_flow21 = renderout["rgb"] * mask_flow2

    if mask_occlusions:
        with torch.no_grad():
            mask_flow2 = renderout["alpha"].unsqueeze(1)
            occl_mask1, occl_mask2 = imgflowarp.get_occlusion_mask(
                mask_flow1, mask_flow2, pred_flow12, pred_flow21
            )
        mask_flow1 = mask_flow1 * occl_mask1.unsqueeze(1)
        mask_flow2 = mask_flow2 * occl_mask2.unsqueeze(1)
        pred_flow12 = pred_flow12 * mask_flow1
        pred_flow21 = pred_flow21 * mask_flow2
    pred_flow12 = pred_flow12.permute(0, 2, 3, 1)[:, :, :, :2]
    pred_flow21 = pred_flow21.permute(0, 2, 3, 1)[:, :, :, :2]
    if orig_img_size is not None:
        pred_flow12 = pred_flow12[:, : orig_img_size[1], : orig_img_size[0]]
        pred_flow21 = pred_flow21[:, : orig_img_size[1], : orig_img_size[0]]
    pred_flows = [pred_flow12, pred_flow21]
    return pred_flows
