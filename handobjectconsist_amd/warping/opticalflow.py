"""Mesh-induced optical flow -- drop-in for meshreg/warping/opticalflow.py.

Same two functions and signatures as the reference (opticalflow.py:10, :51).  The flow is
the per-vertex 2-D displacement ``proj(v_t2) - proj(v_t1)`` painted as a vertex texture
``(dx, dy, 1)`` and rendered with the HIP rasteriser; masks follow the reference's algebra
including its quirks (Q2 un-flipped face_index_map flipped by hand, Q3
ignore list indexes the pre-fill-back faces, Q4 ``mask_flow2`` reset to raw alpha inside
the occlusion block).
"""
from typing import List

import os

import torch

from handobjectconsist_amd import _lib
from handobjectconsist_amd.utils import project, textutils
from handobjectconsist_amd.warping import imgflowarp, pairstep

# Fuse the mask algebra / crop / permute that follows the two renders (opticalflow.py:109-154) into
# three small kernels (mr_flow_mask, mr_occlusion_mask with on-the-fly masked flows,
# mr_flow_finalize_*).  False: the reference's op-by-op structure on torch tensors (same values).
USE_FUSED_EPILOGUE = True


def get_opticalflows(
    verts_cam: List[torch.Tensor],
    faces: torch.Tensor,
    camintrs: List[torch.Tensor],
    neurenderer,
    orig_img_size=None,
    detach_textures: bool = False,
    detach_renders: bool = False,
    ignore_face_idxs=None,
    sparse_flows: bool = False,
):
    """
    Compute optical flow between pairs of meshes (the same mesh at different time steps),
    always comparing to the first mesh (reference opticalflow.py:10-48).

    ``sparse_flows`` (not in the reference): see ``get_opticalflow``.
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
            sparse_flows=sparse_flows,
        )
        all_flows.append(flows)
    return all_flows


# Use the renderer's fused vertex-colour path when it has one and the positions are detached
# (False: always build the face textures and call neurenderer(...) like the reference).
USE_VERTEX_COLOR_RENDER = True


def _render_flow(neurenderer, verts, faces, sample_flows, camintr, detach_textures, detach_renders):
    """neurenderer(verts, faces, batch_vertex_textures(faces, sample_flows), K=..., detach_renders=...)
    (opticalflow.py:103-108)."""
    if _vertex_color_path(neurenderer, detach_renders):
        cols = sample_flows.detach() if detach_textures else sample_flows
        return neurenderer.render_vertex_colors(verts, faces, cols, K=camintr)
    all_textures = textutils.batch_vertex_textures(faces, sample_flows)
    if detach_textures:
        all_textures = all_textures.detach()
    return neurenderer(verts, faces, all_textures, K=camintr, detach_renders=detach_renders)


# One kernel for batch_proj2d of both frames, the two displacement textures and the renderer's
# camera projection of both frames (mr_flow_vertices_forward / _backward) instead of ~150 small
# PyTorch launches; used with the vertex-colour render + fused epilogue.  False: op-by-op as in the
# reference (same values up to fp32 rounding of the 3x3 products).
USE_FUSED_VERTEX_STAGE = True
# ... and render the stacked pair in flow mode (Renderer.render_projected_flow).  False: the full output set of
# render_projected_vertex_colors + a separate mask kernel (same flows, bit for bit).
USE_FLOW_RENDER = True
# ... as one autograd node with a single fused backward launch (_StackedFlowFunction).  False: render, epilogue and
# their backward passes as separate nodes / launches (same values).
USE_STACKED_FLOW_NODE = True
# ... and let the render skip the tiles of the screen that hold no candidate face entirely (MR_FLAG_SPARSE_TILES): the
# occlusion / epilogue pass and the backward consult the render's coverage bytes before every read of a rendered plane.
USE_SPARSE_TILES = True
# ... and leaves per-pixel RECORDS for its backward: the winner's three vertex ids and the three sampling weights their
# colours enter the pixel with (instead of barycentrics + depth, from which the backward had to walk face index ->
# vertex ids -> vertex depths): one load round trip per pixel in mr_render_flow_backward.  Same products, same sums.
USE_PIXEL_RECORDS = True
# tests: allocate the render's output planes filled with NaN / INT_MIN instead of uninitialised, so that any read of a
# pixel the sparse render did not write shows up in the flows or the gradients
DEBUG_POISON_RENDER_OUTPUTS = False
# ... and dispatches workgroups only for the tiles that DO hold candidates: the binning pass compacts them into a list,
# the tile kernel is launched over as many workgroups as the previous call on this (device, batch, raster) had
# list entries (+ margin) -- the kernel reports its list length into a pinned host word, read here without any
# synchronisation; a stale or missing value only changes how the list is split over launches, never the images.
USE_TILE_LIST = True
# ... and, for callers that ask for ``sparse_flows``, runs the occlusion check + flow epilogue over that same list
# (mr_occlusion_flow_tiles): the flows and the occlusion masks are WRITTEN under the covered tiles only, and the list rides
# along with the coverage bytes so that pair_consist (outputs="loss") does the same with its two kernels.
USE_TILE_LIST_WARP = True

_TILE_COUNTS = {}


def _tile_bound(dev, B2, is_):
    """(guess of the tile-list length for mr_render_flow_forward, pinned word the kernel writes the real one to)."""
    key = (dev.index, B2, is_)
    word = _TILE_COUNTS.get(key)
    if word is None:
        word = torch.zeros(1, dtype=torch.int32).pin_memory()
        _TILE_COUNTS[key] = word
    last = int(word[0])  # host memory: no device synchronisation
    return (last + last // 8 + 64 if last > 0 else -1), word


class _FlowVertexStage(torch.autograd.Function):
    """(verts1, verts2, K1, K2; R, t, dist, orig_size) -> (ndc[2B,V,3], cols[2B,V,3]): frame 1 then frame 2
    of every pair, stacked so that both renders of a pair can go out as ONE launch over 2B meshes;
    differentiable w.r.t. the vertices through the two displacement textures only (detach_renders=True)."""

    @staticmethod
    def forward(ctx, verts1, verts2, K1, K2, R, t, dist, orig_size):
        ctx.set_materialize_grads(False)
        v1, v2 = _lib.contig(verts1.detach()), _lib.contig(verts2.detach())
        k1, k2 = _lib.contig(K1.detach()), _lib.contig(K2.detach())
        B, V = v1.shape[:2]
        Rc = _lib.contig(R.detach().reshape(-1, 3, 3))
        tc = _lib.contig(t.detach().reshape(-1, 3))
        dc = _lib.contig(dist.detach().reshape(-1, 5))
        nb = Rc.shape[0]
        if v2.shape != v1.shape or k1.shape != (B, 3, 3) or k2.shape != (B, 3, 3) or nb not in (1, B) \
                or tc.shape[0] != nb or dc.shape[0] != nb:
            raise ValueError("expected vertices [B,V,3], intrinsics [B,3,3] and R / t / dist_coeffs with batch 1 or B")
        ndc = torch.empty((2 * B, V, 3), dtype=torch.float32, device=v1.device)
        cols = torch.empty_like(ndc)
        _lib.call("mr_flow_vertices_forward", _lib.ptr(v1), _lib.ptr(v2), _lib.ptr(k1), _lib.ptr(k2), _lib.ptr(Rc),
                  _lib.ptr(tc), _lib.ptr(dc), int(nb == B and B > 1), float(orig_size), _lib.ptr(ndc[:B]),
                  _lib.ptr(ndc[B:]), _lib.ptr(cols[:B]), _lib.ptr(cols[B:]), B, V, _lib.stream_ptr(v1.device))
        ctx.save_for_backward(v1, v2, k1, k2)
        ctx.mark_non_differentiable(ndc)
        return ndc, cols

    @staticmethod
    def backward(ctx, _g_ndc, g_cols):
        v1, v2, k1, k2 = ctx.saved_tensors
        B, V = v1.shape[:2]
        want1, want2 = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if g_cols is None or not (want1 or want2):
            return (None,) * 8
        gv1 = torch.empty_like(v1) if want1 else None
        gv2 = torch.empty_like(v2) if want2 else None
        g = _lib.contig(g_cols)
        _lib.call("mr_flow_vertices_backward", _lib.ptr(v1), _lib.ptr(v2), _lib.ptr(k1), _lib.ptr(k2), _lib.ptr(g[:B]),
                  _lib.ptr(g[B:]), _lib.ptr(gv1), _lib.ptr(gv2), B, V, _lib.stream_ptr(v1.device))
        return gv1, gv2, None, None, None, None, None, None


class _FlowVertexStageParts(torch.autograd.Function):
    """_FlowVertexStage for meshes handed over as (hand, object) vertex tensors per frame: the concatenation
    ``torch.cat([hand, obj], 1)`` of warpbranch.py:49-55 happens by index inside the kernel (and the split of the
    gradient in its backward) -- two copies and their autograd nodes less per pair."""

    @staticmethod
    def forward(ctx, v1a, v1b, v2a, v2b, K1, K2, R, t, dist, orig_size, hand_face=None, obj_faces=None, clear16=None):
        """``hand_face`` / ``obj_faces`` given: the stacked int32 faces of the pair (``_stack_pair_faces``) come out of the SAME
        launch as a third output (mr_flow_pair_prologue_parts); ``clear16``: ``(address, bytes)`` of a region that launch clears
        (the header of the tile list of the render that follows on this stream + the arrival counters of its binning pass:
        ``mr_render_clear_bytes``, ``_lib.FLAG_TILE_LIST_CLEARED``)."""
        ctx.set_materialize_grads(False)
        parts = [_lib.contig(x.detach()) for x in (v1a, v1b, v2a, v2b)]
        k1, k2 = _lib.contig(K1.detach()), _lib.contig(K2.detach())
        B, Va, Vb = parts[0].shape[0], parts[0].shape[1], parts[1].shape[1]
        Rc = _lib.contig(R.detach().reshape(-1, 3, 3))
        tc = _lib.contig(t.detach().reshape(-1, 3))
        dc = _lib.contig(dist.detach().reshape(-1, 5))
        nb = Rc.shape[0]
        if (parts[2].shape != parts[0].shape or parts[3].shape != parts[1].shape or parts[1].shape[0] != B or k1.shape != (B, 3, 3)
                or k2.shape != (B, 3, 3) or nb not in (1, B) or tc.shape[0] != nb or dc.shape[0] != nb or Vb == 0):
            raise ValueError("expected vertex parts [B,Va,3] / [B,Vb,3], intrinsics [B,3,3] and R / t / dist_coeffs with batch 1 or B")
        ndc = torch.empty((2 * B, Va + Vb, 3), dtype=torch.float32, device=parts[0].device)
        cols = torch.empty_like(ndc)
        head = (*[_lib.ptr(x) for x in parts], Va, Vb, _lib.ptr(k1), _lib.ptr(k2), _lib.ptr(Rc), _lib.ptr(tc), _lib.ptr(dc),
                int(nb == B and B > 1), float(orig_size), _lib.ptr(ndc[:B]), _lib.ptr(ndc[B:]), _lib.ptr(cols[:B]), _lib.ptr(cols[B:]))
        ctx.save_for_backward(*parts, k1, k2)
        ctx.with_faces = obj_faces is not None
        if obj_faces is None:
            _lib.call("mr_flow_vertices_parts_forward", *head, B, _lib.stream_ptr(parts[0].device))
            ctx.mark_non_differentiable(ndc)
            return ndc, cols
        hf, of, batched = _pair_face_parts(hand_face, obj_faces)
        if of.shape[0] != B:
            raise ValueError("expected object faces [B,Fo,3]")
        Fh, Fo = hf.shape[-2], of.shape[1]
        faces2 = torch.empty((2 * B, Fh + Fo, 3), dtype=torch.int32, device=parts[0].device)
        clear_ptr, clear_bytes = clear16 if clear16 is not None else (None, 0)
        _lib.call("mr_flow_pair_prologue_parts", *head, _lib.ptr(hf), int(batched), _lib.ptr(of), _lib.ptr(faces2), Fh, Fo, B,
                  clear_ptr, int(clear_bytes), _lib.stream_ptr(parts[0].device))
        ctx.mark_non_differentiable(ndc, faces2)
        return ndc, cols, faces2

    @staticmethod
    def backward(ctx, _g_ndc, g_cols, _g_faces=None):
        v1a, v1b, v2a, v2b, k1, k2 = ctx.saved_tensors
        B, Va, Vb = v1a.shape[0], v1a.shape[1], v1b.shape[1]
        want = ctx.needs_input_grad[:4]
        if g_cols is None or not any(want):
            return (None,) * 13
        grads = [torch.empty_like(x) if w else None for x, w in zip((v1a, v1b, v2a, v2b), want)]
        g = _lib.contig(g_cols)
        _lib.call("mr_flow_vertices_parts_backward", _lib.ptr(v1a), _lib.ptr(v1b), _lib.ptr(v2a), _lib.ptr(v2b), Va, Vb,
                  _lib.ptr(k1), _lib.ptr(k2), _lib.ptr(g[:B]), _lib.ptr(g[B:]), *[_lib.ptr(x) for x in grads], B,
                  _lib.stream_ptr(v1a.device))
        return tuple(grads) + (None,) * 9


def _pair_face_parts(hand_face, obj_faces):
    """(contiguous int64 hand faces [Fh,3] or [B,Fh,3], object faces [B,Fo,3], hand faces batched?)"""
    hf = _lib.contig(hand_face, torch.int64)
    of = _lib.contig(obj_faces, torch.int64)
    batched = hf.dim() == 3 and hf.shape[0] == of.shape[0] and of.shape[0] > 1
    if hf.dim() == 3 and not batched:
        hf = hf[0]
    return hf, of, batched


def _stack_pair_faces(hand_face, obj_faces, num_hand_verts):
    """int32 [2B, Fh + Fo, 3]: the faces of the concatenated hand + object mesh (object indices offset by the hand's
    vertex count, warpbranch.py:36, 49-55), twice -- what the stacked render of a frame pair takes -- in one launch."""
    hf, of, batched = _pair_face_parts(hand_face, obj_faces)
    B, Fo = of.shape[:2]
    Fh = hf.shape[-2]
    out = torch.empty((2 * B, Fh + Fo, 3), dtype=torch.int32, device=of.device)
    _lib.call("mr_stack_pair_faces", _lib.ptr(hf), int(batched), _lib.ptr(of), int(num_hand_verts), _lib.ptr(out), B, Fh, Fo,
              _lib.stream_ptr(of.device))
    return out


def _vertex_color_path(neurenderer, detach_renders):
    return (USE_VERTEX_COLOR_RENDER and detach_renders and hasattr(neurenderer, "render_vertex_colors")
            and getattr(neurenderer, "no_light", False) and getattr(neurenderer, "camera_mode", "") == "projection")


def _keep_lut(ignore_face_idxs, device):
    """float table over face index + 1 (slot 0 = background): 0 for ignored faces, 1 otherwise."""
    hit = _LUT_BY_ID.get(id(ignore_face_idxs))  # (the trainer hands over the same list object every step)
    if hit is not None and hit[0] is ignore_face_idxs and hit[1] == len(ignore_face_idxs) and hit[2] == device:
        return hit[3]
    key = (tuple(int(i) for i in ignore_face_idxs), str(device))
    lut = _LUT_CACHE.get(key)
    if lut is None:
        ids = torch.as_tensor(list(ignore_face_idxs), dtype=torch.long, device=device)
        n = int(max(int(ids.max().item()) + 2, 2)) if ids.numel() else 2
        lut = torch.ones(n, dtype=torch.float32, device=device)
        if ids.numel():
            lut[ids[ids >= 0] + 1] = 0.0
        _LUT_CACHE[key] = lut
    if isinstance(ignore_face_idxs, (list, tuple)):
        if len(_LUT_BY_ID) > 16:
            _LUT_BY_ID.clear()
        _LUT_BY_ID[id(ignore_face_idxs)] = (ignore_face_idxs, len(ignore_face_idxs), device, lut)
    return lut


def _ignore_mask(face_index_map, ignore_face_idxs):
    """1 where the winning face is not in the ignore list, in IMAGE orientation
    (opticalflow.py:110-116: |fim - ids|.min != 0, then the manual vertical flip).  Done as a
    table lookup over face indices instead of materialising the [B, is, is, 14] difference."""
    lut = _keep_lut(ignore_face_idxs, face_index_map.device)
    idx = (face_index_map.long() + 1).clamp_(max=lut.numel() - 1)
    hi = face_index_map >= (lut.numel() - 1)  # faces beyond the table are never ignored
    keep = torch.where(hi, torch.ones((), device=lut.device), lut[idx])
    return keep.flip(1).unsqueeze(1)


def _flow_mask(renderout, ignore_face_idxs):
    """(alpha > 0.99999) * ignore-mask as one kernel -> [B, is, is] (opticalflow.py:109-117)."""
    alpha = _lib.contig(renderout["alpha"].detach())
    fim = renderout["face_index_map"]
    B, is_ = alpha.shape[0], alpha.shape[1]
    lut = _keep_lut(ignore_face_idxs, alpha.device) if ignore_face_idxs is not None else None
    mask = torch.empty_like(alpha)
    _lib.call("mr_flow_mask", _lib.ptr(alpha), _lib.ptr(fim), _lib.ptr(lut), int(lut.numel()) if lut is not None else 0,
              0.99999, _lib.ptr(mask), B, is_, _lib.stream_ptr(alpha.device))
    return mask, alpha


class _FlowFinalize(torch.autograd.Function):
    """flow[B,H,W,2] = ((rgb * mask_pre) * (mask_x * occl))[:, :2] permuted + cropped
    (opticalflow.py:118, 146-154); differentiable w.r.t. rgb only (the masks carry no gradient)."""

    @staticmethod
    def forward(ctx, rgb, mask_pre, mask_x, occl, height, width):
        rgb_c = _lib.contig(rgb)
        B, _, is_, _ = rgb_c.shape
        flow = torch.empty((B, height, width, 2), dtype=torch.float32, device=rgb_c.device)
        _lib.call("mr_flow_finalize_forward", _lib.ptr(rgb_c), _lib.ptr(mask_pre), _lib.ptr(mask_x), _lib.ptr(occl),
                  _lib.ptr(flow), B, is_, height, width, _lib.stream_ptr(rgb_c.device))
        ctx.save_for_backward(mask_pre, mask_x, occl)
        ctx.dims = (B, is_, height, width)
        return flow

    @staticmethod
    def backward(ctx, grad_flow):
        mask_pre, mask_x, occl = ctx.saved_tensors
        B, is_, height, width = ctx.dims
        g = _lib.contig(grad_flow)
        grad_rgb = torch.empty((B, 3, is_, is_), dtype=torch.float32, device=g.device)
        _lib.call("mr_flow_finalize_backward", _lib.ptr(g), _lib.ptr(mask_pre), _lib.ptr(mask_x), _lib.ptr(occl),
                  _lib.ptr(grad_rgb), B, is_, height, width, _lib.stream_ptr(g.device))
        return grad_rgb, None, None, None, None, None


def _fused_epilogue(ro1, ro2, orig_img_size, ignore_face_idxs):
    """opticalflow.py:109-154 for mask_occlusions=True on the outputs of the two renders."""
    with torch.no_grad():
        m1, _alpha1 = _flow_mask(ro1, ignore_face_idxs)
        m2, alpha2 = _flow_mask(ro2, ignore_face_idxs)
        rgb1, rgb2 = _lib.contig(ro1["rgb"].detach()), _lib.contig(ro2["rgb"].detach())
        B, _, is_, _ = rgb1.shape
        occl1 = torch.empty((B, is_, is_), dtype=torch.float32, device=rgb1.device)
        occl2 = torch.empty_like(occl1)
        # mask_flow2 is the RAW alpha inside the occlusion block (Q4); flows are rgb * mask, on the fly
        _lib.call("mr_occlusion_mask", _lib.ptr(m1), _lib.ptr(alpha2), _lib.ptr(rgb1), _lib.ptr(rgb2), 3 * is_ * is_,
                  _lib.ptr(m1), _lib.ptr(m2), _lib.ptr(occl1), _lib.ptr(occl2), B, is_, is_, 0.03, 0.99999,
                  _lib.stream_ptr(rgb1.device))
    W, H = (orig_img_size[0], orig_img_size[1]) if orig_img_size is not None else (is_, is_)
    W, H = min(int(W), is_), min(int(H), is_)
    flow12 = _FlowFinalize.apply(ro1["rgb"], m1, m1, occl1, H, W)
    flow21 = _FlowFinalize.apply(ro2["rgb"], m2, alpha2, occl2, H, W)
    return [flow12, flow21]


class _FlowFinalizeStacked(torch.autograd.Function):
    """``_FlowFinalize`` of both directions of a pair on the stacked render: rgb[2B,3,is,is] ->
    flow[2B,H,W,2] (first half flow12, second half flow21), one gradient tensor back."""

    @staticmethod
    def forward(ctx, rgb, mask_pre, mask_x1, mask_x2, occl, height, width):
        rgb_c = _lib.contig(rgb)
        B2, _, is_, _ = rgb_c.shape
        B = B2 // 2
        flow = torch.empty((B2, height, width, 2), dtype=torch.float32, device=rgb_c.device)
        for lo, mask_x in ((0, mask_x1), (B, mask_x2)):
            _lib.call("mr_flow_finalize_forward", _lib.ptr(rgb_c[lo:lo + B]), _lib.ptr(mask_pre[lo:lo + B]),
                      _lib.ptr(mask_x), _lib.ptr(occl[lo:lo + B]), _lib.ptr(flow[lo:lo + B]), B, is_, height, width,
                      _lib.stream_ptr(rgb_c.device))
        ctx.save_for_backward(mask_pre, mask_x1, mask_x2, occl)
        ctx.dims = (B, is_, height, width)
        return flow

    @staticmethod
    def backward(ctx, grad_flow):
        mask_pre, mask_x1, mask_x2, occl = ctx.saved_tensors
        B, is_, height, width = ctx.dims
        if grad_flow is None:
            return (None,) * 7
        g = _lib.contig(grad_flow)
        grad_rgb = torch.empty((2 * B, 3, is_, is_), dtype=torch.float32, device=g.device)
        for lo, mask_x in ((0, mask_x1), (B, mask_x2)):
            _lib.call("mr_flow_finalize_backward", _lib.ptr(g[lo:lo + B]), _lib.ptr(mask_pre[lo:lo + B]), _lib.ptr(mask_x),
                      _lib.ptr(occl[lo:lo + B]), _lib.ptr(grad_rgb[lo:lo + B]), B, is_, height, width,
                      _lib.stream_ptr(g.device))
        return grad_rgb, None, None, None, None, None, None


def _fused_epilogue_stacked(ro, orig_img_size, ignore_face_idxs):
    """``_fused_epilogue`` on ONE render of the 2B stacked meshes (frame 1 of every pair, then frame 2)."""
    with torch.no_grad():
        if "mask" in ro:  # the flow-mode render wrote the mask itself
            m, alpha = ro["mask"], ro["alpha"]
        else:
            m, alpha = _flow_mask(ro, ignore_face_idxs)
        rgb = _lib.contig(ro["rgb"].detach())
        B2, _, is_, _ = rgb.shape
        B = B2 // 2
        occl = torch.empty((B2, is_, is_), dtype=torch.float32, device=rgb.device)
        _lib.call("mr_occlusion_mask", _lib.ptr(m[:B]), _lib.ptr(alpha[B:]), _lib.ptr(rgb[:B]), _lib.ptr(rgb[B:]),
                  3 * is_ * is_, _lib.ptr(m[:B]), _lib.ptr(m[B:]), _lib.ptr(occl[:B]), _lib.ptr(occl[B:]), B, is_, is_,
                  0.03, 0.99999, _lib.stream_ptr(rgb.device))
    W, H = (orig_img_size[0], orig_img_size[1]) if orig_img_size is not None else (is_, is_)
    W, H = min(int(W), is_), min(int(H), is_)
    flows = _FlowFinalizeStacked.apply(ro["rgb"], m, m[:B], alpha[B:], occl, H, W)
    # two views of one tensor: imgflowarp.pair_consist recognises them and differentiates the stack
    return [flows[:B], flows[B:]]


def _render_stacked_flow(ndc, faces2, cols, lut, fill_back, image_size, near, far, eps, background_color, want_grad,
                         cleared_work=None):
    """The flow-mode render of the 2B stacked meshes of a frame pair (mr_render_flow_forward): what both training nodes
    (_StackedFlowFunction, _FlowPairLossFunction) start with.  Returns the buffers by name.
    ``cleared_work``: the render's workspace, allocated by the caller, the header of its tile list cleared on this stream
    (the pair prologue does that): the render then runs its per-face pass inside the binning pass."""
    from handobjectconsist_amd.neurender import rasterize

    _lib.check_cuda(ndc, faces2, cols, lut)
    if not (float(eps) >= 1e-6):
        raise ValueError("vertex-colour rendering needs eps >= 1e-6")
    verts, fidx, c = _lib.contig(ndc.detach()), faces2, _lib.contig(cols.detach())
    dev = verts.device
    B2, V = verts.shape[:2]
    F0, is_ = fidx.shape[1], int(image_size)
    if (fidx.dtype != torch.int32 or not fidx.is_contiguous() or B2 % 2 or fidx.shape != (B2, F0, 3)
            or verts.shape != (B2, V, 3) or c.shape != (B2, V, 3)):
        raise ValueError("expected stacked vertices / colours [2B,V,3] and contiguous int32 faces [2B,F,3]")
    f32 = dict(dtype=torch.float32, device=dev)
    bg, bg_stride = rasterize._background_tensor(background_color, dev, B2)
    if DEBUG_POISON_RENDER_OUTPUTS:
        new_f = lambda *shape: torch.full(shape, float("nan"), **f32)
        new_i = lambda *shape: torch.full(shape, -2 ** 31, dtype=torch.int32, device=dev)
    else:
        new_f = lambda *shape: torch.empty(shape, **f32)
        new_i = lambda *shape: torch.empty(shape, dtype=torch.int32, device=dev)
    rgb = new_f(B2, 3, is_, is_)
    alpha, mask = new_f(B2, is_, is_), new_f(B2, is_, is_)
    # the backward's inputs, valid at covered pixels only: sampling weights + vertex ids, or barycentrics + depth
    depth = None if USE_PIXEL_RECORDS else new_f(B2, is_, is_)
    vid = new_i(B2, is_, is_, 3) if USE_PIXEL_RECORDS else None
    wmap = new_f(B2, is_, is_, 3)
    fim = new_i(B2, is_, is_)
    tile_hit = torch.empty((B2, (is_ + 7) // 8, (is_ + 31) // 32, 4), dtype=torch.uint8, device=dev)
    F = 2 * F0 if fill_back else F0
    wbytes = int(_lib.load().mr_render_workspace_bytes(B2, F, is_))
    if cleared_work is not None and (cleared_work.numel() < wbytes or cleared_work.device != dev):
        raise ValueError("the render's workspace is too small or on another device")
    work = cleared_work if cleared_work is not None else torch.empty((max(wbytes, 8),), dtype=torch.uint8, device=dev)
    st = _lib.stream_ptr(dev)
    bound, count_word = _tile_bound(dev, B2, is_) if (USE_SPARSE_TILES and USE_TILE_LIST) else (0, None)
    render_flags = ((_lib.FLAG_SPARSE_TILES if USE_SPARSE_TILES else 0) | (_lib.FLAG_TILE_LIST_CLEARED if cleared_work is not None else 0)
                    | _FWD_DBG_FLAGS)
    # the backward's output buffer is cleared by the render's binning pass on its way (its own clearing would be a
    # launch on the backward pass's critical path); a second backward through this node clears its own
    grad_buf = torch.empty((B2, V, 3), **f32) if want_grad else None
    _lib.call("mr_render_flow_forward", _lib.ptr(verts), _lib.ptr(fidx), _lib.ptr(c), _lib.ptr(bg), bg_stride,
              _lib.ptr(lut), int(lut.numel()) if lut is not None else 0, 0.99999, _lib.ptr(rgb), _lib.ptr(alpha),
              _lib.ptr(mask), _lib.ptr(depth), _lib.ptr(wmap), _lib.ptr(fim), _lib.ptr(tile_hit), _lib.ptr(work), wbytes,
              B2, V, F0, int(bool(fill_back)), is_, float(near), float(far), float(eps),
              render_flags, _lib.ptr(vid), bound, _lib.ptr(count_word), _lib.ptr(grad_buf),
              int(grad_buf.numel()) if grad_buf is not None else 0, textutils.texel_layout_code(), st)
    return dict(verts=verts, fidx=fidx, rgb=rgb, alpha=alpha, mask=mask, depth=depth, vid=vid, wmap=wmap, fim=fim,
                tile_hit=tile_hit, work=work, bound=bound, grad_buf=grad_buf, new_f=new_f, f32=f32, dev=dev, st=st, B2=B2, V=V,
                F=F, F0=F0, is_=is_)


class _StackedFlowFunction(torch.autograd.Function):
    """The whole training-path body of ``get_opticalflow`` after the vertex stage as ONE autograd node:
    (ndc[2B,V,3], faces[2B,F0,3] int32, cols[2B,V,3]) -> flows[2B,H,W,2] (first half flow12, second half flow21).

    forward:  flow-mode render of the 2B stacked meshes (mr_render_flow_forward: displacement planes, alpha, flow
              mask, face index, weights / depth at covered pixels, per-tile coverage bytes), then occlusion check
              (SURVEY Q4 masks) + crop / permute / mask products of both directions in one pass (mr_occlusion_flow);
    backward: ONE launch (mr_render_flow_backward): the adjoint of the epilogue is applied on the fly to the
              flow-space gradient, the colour-space gradient [2B,3,is,is] is never materialised, empty tiles are
              skipped on the coverage bytes.
    Differentiable w.r.t. ``cols`` only (detach_renders=True).
    ``sparse``: the occlusion / epilogue pass runs over the render's tile list and writes ``flow`` / ``occl`` under the
    covered tiles only (mr_occlusion_flow_tiles); the list goes back to ``get_opticalflow`` in ``tile_out`` (a list the caller
    owns: nothing is kept on the class, calls may nest or run from several threads)."""

    @staticmethod
    def forward(ctx, ndc, faces2, cols, lut, fill_back, image_size, near, far, eps, background_color, height, width,
                sparse=False, tile_out=None):
        ctx.set_materialize_grads(False)
        r = _render_stacked_flow(ndc, faces2, cols, lut, fill_back, image_size, near, far, eps, background_color,
                                 ctx.needs_input_grad[2])
        verts, fidx, rgb, alpha, mask, depth, vid, wmap, fim, tile_hit, work, bound, grad_buf = (
            r["verts"], r["fidx"], r["rgb"], r["alpha"], r["mask"], r["depth"], r["vid"], r["wmap"], r["fim"], r["tile_hit"],
            r["work"], r["bound"], r["grad_buf"])
        new_f, f32, dev, st = r["new_f"], r["f32"], r["dev"], r["st"]
        B2, V, B, F, is_ = r["B2"], r["V"], r["B2"] // 2, r["F"], r["is_"]
        # the render's tile list, for callers that accept flows defined under the covered tiles only
        tiles = None
        if sparse and USE_TILE_LIST_WARP and bound != 0:
            where = _lib.tile_list(work, B2, F, is_)
            if where is not None:
                tiles = (where[0], where[1], where[2], int(bound), work)  # (`work` rides along: the list lives in it)
        if tile_out is not None:
            tile_out.append(tiles)
        occl = new_f(B2, is_, is_) if tiles else torch.empty((B2, is_, is_), **f32)
        flow = new_f(B2, height, width, 2) if tiles else torch.empty((B2, height, width, 2), **f32)
        # occlusion check + crop / permute / mask products of both directions in one pass.  mask_flow2 is the RAW
        # alpha inside the occlusion block and afterwards (Q4); the masked flows rgb * mask are formed on the fly
        if tiles:
            _lib.call("mr_occlusion_flow_tiles", _lib.ptr(mask[:B]), _lib.ptr(alpha[B:]), _lib.ptr(rgb[:B]), _lib.ptr(rgb[B:]),
                      3 * is_ * is_, _lib.ptr(mask[:B]), _lib.ptr(mask[B:]), _lib.ptr(occl[:B]), _lib.ptr(occl[B:]),
                      _lib.ptr(flow[:B]), _lib.ptr(flow[B:]), _lib.ptr(tile_hit[:B]), _lib.ptr(tile_hit[B:]), B, is_, height,
                      width, 0.03, 0.99999, tiles[0], tiles[1], tiles[2], tiles[3], st)
        else:
            _lib.call("mr_occlusion_flow", _lib.ptr(mask[:B]), _lib.ptr(alpha[B:]), _lib.ptr(rgb[:B]), _lib.ptr(rgb[B:]),
                      3 * is_ * is_, _lib.ptr(mask[:B]), _lib.ptr(mask[B:]), _lib.ptr(occl[:B]), _lib.ptr(occl[B:]),
                      _lib.ptr(flow[:B]), _lib.ptr(flow[B:]), _lib.ptr(tile_hit[:B]), _lib.ptr(tile_hit[B:]), B, is_, is_, height,
                      width, 0.03, 0.99999, st)
        ctx.tiles = tiles
        ctx.cfg = (is_, float(eps), bool(fill_back), height, width)
        ctx.save_for_backward(verts, fidx, fim, tile_hit, wmap, depth if depth is not None else vid, mask, alpha, occl)
        ctx.records = depth is None
        ctx.grad_buf = grad_buf
        ctx.mark_non_differentiable(tile_hit)
        return flow, tile_hit

    @staticmethod
    def backward(ctx, grad_flow, _grad_hit=None):
        verts, fidx, fim, tile_hit, wmap, depth_or_vid, mask, alpha, occl = ctx.saved_tensors
        depth, vid = (None, depth_or_vid) if ctx.records else (depth_or_vid, None)
        is_, eps, fill_back, height, width = ctx.cfg
        if grad_flow is None or not ctx.needs_input_grad[2]:
            return (None,) * 14
        B2, V = verts.shape[:2]
        B = B2 // 2
        g = _lib.contig(grad_flow)
        # an upper bound of |grad_flow| per image, if the producer of the gradient left one (the stacked pair loss does)
        note = getattr(grad_flow, "_hoc_grad_bound", None)
        bound = note[0] if (note is not None and g is grad_flow and note[1] == grad_flow._version
                            and tuple(note[0].shape) == (B2,) and note[0].device == g.device) else None
        grad_cols, ctx.grad_buf = ctx.grad_buf, None
        zeroed = grad_cols is not None
        if not zeroed:
            grad_cols = torch.empty((B2, V, 3), dtype=torch.float32, device=verts.device)
        _lib.call("mr_render_flow_backward", _lib.ptr(verts), _lib.ptr(fidx), _lib.ptr(fim), _lib.ptr(tile_hit), _lib.ptr(wmap),
                  _lib.ptr(depth), None, _lib.ptr(g), _lib.ptr(mask), _lib.ptr(mask[:B]), _lib.ptr(alpha[B:]), B,
                  _lib.ptr(occl), height, width, _lib.ptr(grad_cols), B2, V, int(fidx.shape[1]), int(fill_back), is_, eps,
                  _lib.FLAG_OUTPUT_ZEROED if zeroed else 0, _lib.ptr(vid), textutils.texel_layout_code(), _lib.ptr(bound),
                  _lib.stream_ptr(verts.device))
        return (None, None, grad_cols) + (None,) * 11


# The consistency term of a frame pair as ONE autograd node (warpbranch's "loss" mode): flow render, then occlusion check +
# flow epilogue + pair loss in one pass over the render's tile list (mr_flow_pair_forward_tiles), and ONE backward launch in
# which the pair loss's backward, the epilogue's adjoint and the scatter to the vertex colours run per covered tile
# (mr_flow_pair_backward_tiles: the flow gradient never exists as a tensor).  False: get_opticalflow -> pair_consist, the
# same kernels' arithmetic in five launches (same losses bit for bit, gradients to fp32 rounding).
USE_FUSED_PAIR_NODE = True
# ... with the pair loss's gradient formed by the forward launch, where its taps and masks already sit in registers
# (mr_flow_pair_forward_grad_tiles: 8 B per covered pixel more to write), the backward launch being the scatter alone on
# (that gradient) x (grad_loss / count) (mr_flow_pair_backward_unit_tiles): no image, mask or flow is read twice.
# False: mr_flow_pair_forward_tiles + mr_flow_pair_backward_tiles (the backward recomputes the taps).
USE_UNIT_GRADIENT = True
# ... and with the backward's workgroups handed out over the covered-tile lists the forward's finalize launch compacts (ABI 7):
# workgroups per image in proportion to its covered tiles.  False: a fixed number per image, each listing the image's tiles.
USE_SCATTER_WORK = os.environ.get("HOC_SCATTER_WORK", "1") != "0"
_FWD_DBG_FLAGS = int(os.environ.get("HOC_FWD_DBG", "0")) << 8  # profiling switches of the forward kernels (csrc/raster_fwd.hip: dbg)
_PAIR_STEP_FLAGS = int(os.environ.get("HOC_PAIR_STEP_FLAGS", "0")) & 0xfe  # MrPairStep.flags, e.g. 2 = MR_PAIR_STEP_SEPARATE_LAUNCHES (A / B runs)
# ... and with the render's per-face pass folded into its binning pass (the pair prologue clears the tile list's header, which
# the per-face pass's first thread does otherwise): one launch and one dependent round trip less per pair.
USE_FUSED_RECORDS = True
# ... and with all of it behind two struct calls and one autograd node (ABI 8, warping/pairstep.py): the host side of the pair.
# False: the node pair below (_FlowVertexStageParts + _FlowPairLossFunction), five calls -- same kernels, same values.
USE_PAIR_STEP = os.environ.get("HOC_PAIR_STEP", "1") != "0"
# fourth element of a flows tensor's coverage note when the render's tile list is NOT available to later passes
COVERAGE_ONLY = ("coverage-only",)


class _FlowPairLossFunction(torch.autograd.Function):
    """(ndc[2B,V,3], faces[2B,F0,3] int32, cols[2B,V,3]; image_ref, image [B,3,H,W], jitter masks [B,Cj,H,W]) ->
    (loss_fwd[B], loss_bwd[B], loss_bwd + loss_fwd, flows[2B,H,W,2], tile_hit): opticalflow.py:98-154 + imgflowarp.py:58-115 +
    pyramidloss.py:56-62 + lossutils.py:1-8 for one frame pair.  Differentiable w.r.t. ``cols`` only (the training
    setting: detach_renders=True, images are data).  ``flows`` are defined under the covered tiles only; the render's tile
    list goes back to the caller in ``tile_out`` (a list the caller owns)."""

    @staticmethod
    def forward(ctx, ndc, faces2, cols, lut, fill_back, image_size, near, far, eps, background_color, height, width,
                image_ref, image, jitter_ref, jitter, thresh, cleared_work=None, tile_out=None):
        ctx.set_materialize_grads(False)
        _lib.check_cuda(image_ref, image, jitter_ref, jitter)
        r = _render_stacked_flow(ndc, faces2, cols, lut, fill_back, image_size, near, far, eps, background_color,
                                 ctx.needs_input_grad[2], cleared_work)
        B2, B, is_, dev, st, new_f, f32 = r["B2"], r["B2"] // 2, r["is_"], r["dev"], r["st"], r["new_f"], r["f32"]
        where = _lib.tile_list(r["work"], B2, r["F"], is_) if r["bound"] != 0 else None
        if where is None or r["vid"] is None:
            raise RuntimeError("the fused pair node needs the render's tile list and per-pixel records")
        im_ref, im, jm_ref, jm = (_lib.contig(x) for x in (image_ref, image, jitter_ref, jitter))
        Cj = jm.shape[1]
        if (im.shape != (B, 3, height, width) or im_ref.shape != im.shape or Cj not in (1, 3)
                or jm.shape != (B, Cj, height, width) or jm_ref.shape != jm.shape or width < 2):
            raise ValueError("images must be [B,3,H,W] and jitter masks [B,1 or 3,H,W] of the flows' size")
        rgb, alpha, mask, tile_hit = r["rgb"], r["alpha"], r["mask"], r["tile_hit"]
        occl, flow = new_f(B2, is_, is_), new_f(B2, height, width, 2)
        wbytes = int(_lib.load().mr_pair_consist_tiles_workspace_bytes(B, is_))
        work = torch.empty((max(wbytes, 16),), dtype=torch.uint8, device=dev)
        sums, loss_fwd, loss_bwd = torch.empty((B, 4), **f32), torch.empty((B,), **f32), torch.empty((B,), **f32)
        unit = ctx.needs_input_grad[2] and USE_UNIT_GRADIENT
        args = (_lib.ptr(mask[:B]), _lib.ptr(alpha[B:]), _lib.ptr(rgb[:B]), _lib.ptr(rgb[B:]),
                3 * is_ * is_, _lib.ptr(mask[:B]), _lib.ptr(mask[B:]), _lib.ptr(occl[:B]), _lib.ptr(occl[B:]),
                _lib.ptr(flow[:B]), _lib.ptr(flow[B:]), _lib.ptr(tile_hit[:B]), _lib.ptr(tile_hit[B:]), _lib.ptr(im_ref),
                _lib.ptr(im), _lib.ptr(jm_ref), _lib.ptr(jm), Cj, _lib.ptr(work), wbytes, _lib.ptr(sums), _lib.ptr(loss_fwd),
                _lib.ptr(loss_bwd), B, is_, height, width, 0.03, 0.99999, float(thresh), where[0], where[1], where[2],
                int(r["bound"]))
        if unit:
            unit_grad, unit_max, loss_sum = new_f(B2, height, width, 2), torch.empty((B2,), **f32), torch.empty((B,), **f32)
            # the images' covered-tile lists: the finalize launch writes them, the backward hands out its workgroups over them
            scatter_work = (torch.empty((int(_lib.load().mr_flow_pair_scatter_work_bytes(B, is_)),), dtype=torch.uint8, device=dev)
                            if USE_SCATTER_WORK else None)
            _lib.call("mr_flow_pair_forward_grad_tiles", *args, _lib.ptr(unit_grad), _lib.ptr(unit_max), _lib.ptr(loss_sum),
                      _lib.ptr(scatter_work), st)
        else:
            _lib.call("mr_flow_pair_forward_tiles", *args, st)
            loss_sum = loss_bwd + loss_fwd
        # (the flows are defined under the covered tiles only: the list rides along with them, as for get_opticalflow(sparse_flows=True))
        if tile_out is not None:
            tile_out.append((where[0], where[1], where[2], int(r["bound"]), r["work"]))
        ctx.cfg = (is_, float(eps), bool(fill_back), height, width, float(thresh), int(r["F0"]), int(r["V"]))
        ctx.unit = unit
        if unit:
            ctx.save_for_backward(r["fim"], tile_hit, r["wmap"], r["vid"], unit_grad, unit_max, sums, scatter_work)
        else:
            ctx.save_for_backward(r["fim"], tile_hit, r["wmap"], r["vid"], mask, alpha, occl, flow, im_ref, im, jm_ref, jm, sums)
        ctx.grad_buf = r["grad_buf"]
        ctx.mark_non_differentiable(flow, tile_hit)
        return loss_fwd, loss_bwd, loss_sum, flow, tile_hit

    @staticmethod
    def backward(ctx, g_fwd, g_bwd, g_sum=None, _g_flow=None, _g_hit=None):
        is_, eps, fill_back, height, width, thresh, F0, V = ctx.cfg
        if g_sum is not None:  # d/d(loss_bwd + loss_fwd) goes to both terms (the same tensor for both when only the sum is used)
            g_fwd = g_sum if g_fwd is None else g_fwd + g_sum
            g_bwd = g_sum if g_bwd is None else g_bwd + g_sum
        if not ctx.needs_input_grad[2] or (g_fwd is None and g_bwd is None):
            return (None,) * 19
        fim = ctx.saved_tensors[0]
        B2 = fim.shape[0]
        B, dev = B2 // 2, fim.device
        if g_fwd is None:
            g_fwd = torch.zeros((B,), dtype=torch.float32, device=dev)
        g_fwd, g_bwd = _lib.contig(g_fwd), (_lib.contig(g_bwd) if g_bwd is not None else None)
        grad_cols, ctx.grad_buf = ctx.grad_buf, None
        zeroed = grad_cols is not None
        if not zeroed:
            grad_cols = torch.empty((B2, V, 3), dtype=torch.float32, device=dev)
        if ctx.unit:
            fim, tile_hit, wmap, vid, unit_grad, unit_max, sums, scatter_work = ctx.saved_tensors
            _lib.call("mr_flow_pair_backward_unit_tiles", _lib.ptr(fim), _lib.ptr(tile_hit), _lib.ptr(wmap), _lib.ptr(vid),
                      _lib.ptr(unit_grad), _lib.ptr(unit_max), _lib.ptr(sums), _lib.ptr(g_fwd), _lib.ptr(g_bwd), height, width,
                      _lib.ptr(grad_cols), B2, V, F0, int(fill_back), is_, eps, _lib.FLAG_OUTPUT_ZEROED if zeroed else 0,
                      textutils.texel_layout_code(), _lib.ptr(scatter_work), _lib.stream_ptr(dev))
            return (None, None, grad_cols) + (None,) * 16
        fim, tile_hit, wmap, vid, mask, alpha, occl, flow, im_ref, im, jm_ref, jm, sums = ctx.saved_tensors
        # scratch of the launch: the masked flow gradient of a workgroup's tiles between its two passes
        scratch = (torch.full((B2, height, width, 2), float("nan"), dtype=torch.float32, device=dev) if DEBUG_POISON_RENDER_OUTPUTS
                   else torch.empty((B2, height, width, 2), dtype=torch.float32, device=dev))
        _lib.call("mr_flow_pair_backward_tiles", _lib.ptr(fim), _lib.ptr(tile_hit), _lib.ptr(wmap), _lib.ptr(vid), _lib.ptr(flow),
                  _lib.ptr(im_ref), _lib.ptr(im), _lib.ptr(jm_ref), _lib.ptr(jm), int(jm.shape[1]), _lib.ptr(sums), _lib.ptr(g_fwd),
                  _lib.ptr(g_bwd), _lib.ptr(mask), _lib.ptr(mask[:B]), _lib.ptr(alpha[B:]), _lib.ptr(occl), _lib.ptr(scratch),
                  height, width, _lib.ptr(grad_cols), B2, V, F0, int(fill_back), is_, eps, thresh,
                  _lib.FLAG_OUTPUT_ZEROED if zeroed else 0, textutils.texel_layout_code(), _lib.stream_ptr(dev))
        return (None, None, grad_cols) + (None,) * 16


def dense_flows(pair_flows):
    """The two flows of a pair with zeros wherever nothing was rendered, as the reference returns them (opticalflow.py:151-156).
    For flows that ``flow_pair_loss`` / ``get_opticalflow(..., sparse_flows=True)`` wrote under their renders' covered tiles
    only (unspecified memory elsewhere: warpbranch's "loss" mode): a masked COPY, made on request -- logging, visualisation,
    reductions over whole flows.  Flows that are dense already come back as they are."""
    base = getattr(pair_flows[0], "_base", None)
    note = getattr(base, "_hoc_coverage", None) if base is not None else None
    if note is None or len(note) < 4 or note[3] is None:
        return list(pair_flows)
    tile_hit, is_, version = note[0], int(note[1]), note[2]
    if version != base._version:
        raise RuntimeError("the flows were written in place after their renders: their coverage is no longer known")
    B2, H, W, _ = base.shape
    # one 4-byte coverage word per tile of 8 raster rows x 32 columns; a tile whose word is non-zero was written completely.
    # Raster row r is image row is - 1 - r (the renderer's vertical flip).
    on = tile_hit.view(torch.int32)[..., 0] != 0
    on = on.repeat_interleave(8, 1)[:, :is_].flip(1).repeat_interleave(32, 2)[:, :H, :W]
    dense = torch.where(on[..., None], base, torch.zeros((), dtype=base.dtype, device=base.device))
    return [dense[:B2 // 2], dense[B2 // 2:]]


def flow_pair_loss(verts_cam, faces, camintrs, neurenderer, orig_img_size, image_ref, image, jitter_mask_ref, jitter_mask,
                   ignore_face_idxs=None, with_sum=False, with_mean=None):
    """``get_opticalflow(verts_cam, ..., detach_textures=False, detach_renders=True)`` followed by
    ``pair_consist(flows, image_ref, image, jitter_mask_ref, jitter_mask, PyramidCriterion("l1"))`` for ONE frame pair, as a
    single fused node (no counterpart function in the reference: opticalflow.py:51-156 + imgflowarp.py:58-115 composed).

    ``verts_cam`` = two [B,V,3] tensors and ``faces`` = [B,F,3], as for ``get_opticalflow`` -- or, to spare the copies of
    warpbranch.py:49-55, two ``(hand [B,Vh,3], object [B,Vo,3])`` tuples and ``faces`` = ``(hand_faces [Fh,3] or [B,Fh,3],
    object_faces [B,Fo,3])`` (object indices WITHOUT the hand offset): concatenation and offset then happen inside the kernels.

    Returns ``(loss_fwd[B], loss_bwd[B], [flow12, flow21])`` -- ``pair_consist``'s ``warp_loss`` is ``loss_fwd`` (+
    ``loss_bwd`` with ``use_backward``); the flows are defined under their renders' covered tiles only -- or ``None`` when the
    fused node does not apply (renderer settings, raster size, tensors off the GPU): callers then compose the two functions.
    ``with_sum``: a fourth element, ``loss_bwd + loss_fwd`` as the node's own output (the finalize launch writes it: one
    element-wise launch less each way for callers that want the sum).  ``with_mean`` ("sum" or "fwd"): one more element, the
    mean over the batch of ``loss_bwd + loss_fwd`` / of ``loss_fwd`` (warpbranch.py:87-88 for one pair) -- the node's own
    output where the pair goes through ``pairstep`` (ABI 8: (hand, object) parts whose vertices want a gradient), a
    ``torch.mean`` otherwise."""
    parts = isinstance(verts_cam[0], (tuple, list))  # (hand, object) vertex tensors per frame + (hand, object) faces
    if parts:
        (h1, o1), (h2, o2) = verts_cam
        hand_face, obj_faces = faces
        tensors_ok = (all(x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 for x in (h1, o1, h2, o2))
                      and h2.shape == h1.shape and o2.shape == o1.shape and o1.shape[0] == h1.shape[0] and obj_faces.is_cuda
                      and hand_face.is_cuda and obj_faces.dim() == 3 and obj_faces.shape[0] == h1.shape[0])
        B, V, dev = h1.shape[0], h1.shape[1] + (o1.shape[1] if tensors_ok else 0), h1.device
        num_faces0 = (hand_face.shape[-2] + obj_faces.shape[1]) if tensors_ok else 0
    else:
        v1, v2 = verts_cam
        tensors_ok = v1.is_cuda and v1.dtype == torch.float32 and v2.shape == v1.shape and v1.dim() == 3
        B, V, dev = v1.shape[0], v1.shape[1], v1.device
        num_faces0 = faces.shape[1]
    if not (USE_FUSED_PAIR_NODE and USE_FUSED_VERTEX_STAGE and USE_FUSED_EPILOGUE and USE_SPARSE_TILES and USE_TILE_LIST
            and USE_PIXEL_RECORDS and USE_TILE_LIST_WARP and _vertex_color_path(neurenderer, True)
            and hasattr(neurenderer, "render_projected_vertex_colors") and tensors_ok
            and _stacked_flow_node_ok(neurenderer, V) and image.is_cuda
            and image.dtype == torch.float32 and image.dim() == 4 and image.shape[1] == 3 and image.shape[-1] >= 2
            and jitter_mask.dim() == 4 and jitter_mask.shape[1] in (1, 3)
            # (the node differentiates w.r.t. the vertices only: images that want a gradient take the composed path)
            and not (image.requires_grad or image_ref.requires_grad or jitter_mask.requires_grad or jitter_mask_ref.requires_grad)):
        return None
    is_ = int(neurenderer.image_size)
    W, H = (orig_img_size[0], orig_img_size[1]) if orig_img_size is not None else (is_, is_)
    H, W = min(int(H), is_), min(int(W), is_)
    if tuple(image.shape[2:]) != (H, W) or image_ref.shape != image.shape:
        return None
    if (parts and USE_PAIR_STEP and USE_UNIT_GRADIENT and USE_SCATTER_WORK and USE_FUSED_RECORDS and torch.is_grad_enabled()
            and (h1.requires_grad or o1.requires_grad or h2.requires_grad or o2.requires_grad)):
        lut = _keep_lut(ignore_face_idxs, dev) if ignore_face_idxs is not None else None
        res = pairstep.pair_step((h1, o1), (h2, o2), hand_face, obj_faces, camintrs[0].to(dev), camintrs[1].to(dev), neurenderer,
                                 is_, H, W, image_ref, image, jitter_mask_ref, jitter_mask, lut, mean_of_fwd_only=(with_mean == "fwd"),
                                 poison=DEBUG_POISON_RENDER_OUTPUTS, flags=_FWD_DBG_FLAGS | _PAIR_STEP_FLAGS)
        if res is not None:
            mean, loss_sum, loss_fwd, loss_bwd, flows, tile_hit = res
            # (the render's tile list lives in the plan's scratch, which the next call reuses: the note carries the coverage only)
            flows._hoc_coverage = (tile_hit, is_, flows._version, COVERAGE_ONLY)
            out = [loss_fwd, loss_bwd, [flows[:B], flows[B:]]]
            if with_sum:
                out.append(loss_sum)
            if with_mean:
                out.append(mean)
            return tuple(out)
    F = num_faces0 * (2 if neurenderer.fill_back else 1)
    if not _lib.has_tile_list(2 * B, F, is_):
        return None
    cam = (camintrs[0].to(dev), camintrs[1].to(dev), neurenderer.R.to(dev), neurenderer.t.to(dev), neurenderer.dist_coeffs.to(dev),
           neurenderer.orig_size)
    cleared_work = None
    if parts:
        # the render's workspace is allocated here so that the prologue launch can clear the header of its tile list: the
        # render's per-face pass (whose first thread does that otherwise) then runs inside its binning pass
        clear16 = None
        if USE_FUSED_RECORDS and USE_SPARSE_TILES and USE_TILE_LIST:
            wbytes = int(_lib.load().mr_render_workspace_bytes(2 * B, F, is_))
            cleared_work = torch.empty((max(wbytes, 8),), dtype=torch.uint8, device=dev)
            where = _lib.tile_list(cleared_work, 2 * B, F, is_)
            if where is None:
                cleared_work = None
            else:
                clear16 = (where[0], int(_lib.load().mr_render_clear_bytes(2 * B, F, is_)))
        ndc, cols, faces2 = _FlowVertexStageParts.apply(h1, o1, h2, o2, *cam, hand_face, obj_faces, clear16)  # (+ the stacked faces)
    else:
        ndc, cols = _FlowVertexStage.apply(v1, v2, *cam)
        faces2 = _stacked_faces(faces)
    lut = _keep_lut(ignore_face_idxs, dev) if ignore_face_idxs is not None else None
    loss_fwd, loss_bwd, loss_sum, flows, tile_hit = _FlowPairLossFunction.apply(
        ndc, faces2, cols, lut, neurenderer.fill_back, is_, neurenderer.near, neurenderer.far,
        neurenderer.rasterizer_eps, neurenderer.background_color, H, W, image_ref, image, jitter_mask_ref, jitter_mask, 0.99999,
        cleared_work, tile_out := [])
    tiles = tile_out[0]
    flows._hoc_coverage = (tile_hit, is_, flows._version, tiles)
    out = [loss_fwd, loss_bwd, [flows[:B], flows[B:]]]
    if with_sum:
        out.append(loss_sum)
    if with_mean:
        out.append((loss_fwd if with_mean == "fwd" else loss_sum).mean())
    return tuple(out)


def _stacked_flow_node_ok(neurenderer, num_verts):
    """mr_render_flow_backward reads 4-pixel groups with 16-byte loads and keeps a [V,3] table in LDS."""
    is_ = int(neurenderer.image_size)
    return (USE_FLOW_RENDER and USE_STACKED_FLOW_NODE and not neurenderer.anti_aliasing and is_ % 4 == 0
            and ((is_ + 31) // 32) * ((is_ + 7) // 8) <= 4096 and num_verts <= 2560)


_FACES2_CACHE = {}


def _stacked_faces(faces):
    """int32 ``cat([faces, faces])`` for the 2B stacked render, cached on the tensor's identity/version."""
    key = (faces.data_ptr(), faces._version, tuple(faces.shape), faces.dtype, str(faces.device))
    hit = _FACES2_CACHE.get("f")
    if hit is None or hit[0] != key:
        f32 = faces.detach().to(torch.int32)
        hit = (key, torch.cat([f32, f32], 0).contiguous(), faces)  # keeps `faces` alive: data_ptr stays unique
        _FACES2_CACHE["f"] = hit
    return hit[1]


_LUT_CACHE = {}
_LUT_BY_ID = {}


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
    sparse_flows: bool = False,
):
    """
    Compute optical flow in image space given the displacement of the vertices in
    verts_cam (reference opticalflow.py:51-156).

    ``sparse_flows`` (not in the reference; default off = fully defined tensors): the caller promises to read the
    returned flows only where the coverage bytes that ride on them (``flows[0]._base._hoc_coverage``) are non-zero --
    ``pair_consist(..., outputs="loss")`` does.  The training path then neither computes nor WRITES anything under the
    five sixths of the screen no mesh touches (the flows are exactly zero there; with this flag those zeros are simply
    not stored).  Ignored on every path but the stacked training node.

    Returns:
        [pred_flow12, pred_flow21], each [batch_size, H, W, 2] in pixel units.
    """
    if (USE_FUSED_VERTEX_STAGE and USE_FUSED_EPILOGUE and mask_occlusions and _vertex_color_path(neurenderer, detach_renders)
            and hasattr(neurenderer, "render_projected_vertex_colors") and verts_cam[0].is_cuda
            and verts_cam[0].dtype == torch.float32 and verts_cam[1].shape == verts_cam[0].shape):
        dev = verts_cam[0].device
        ndc, cols = _FlowVertexStage.apply(
            verts_cam[0], verts_cam[1], camintrs[0].to(dev), camintrs[1].to(dev), neurenderer.R.to(dev),
            neurenderer.t.to(dev), neurenderer.dist_coeffs.to(dev), neurenderer.orig_size)
        B = verts_cam[0].shape[0]
        if detach_textures:  # only the first texture set is detached (opticalflow.py:100-102 vs :123)
            ro1 = neurenderer.render_projected_vertex_colors(ndc[:B], faces, cols[:B].detach())
            ro2 = neurenderer.render_projected_vertex_colors(ndc[B:], faces, cols[B:])
            return _fused_epilogue(ro1, ro2, orig_img_size, ignore_face_idxs)
        if _stacked_flow_node_ok(neurenderer, ndc.shape[1]):
            is_ = int(neurenderer.image_size)
            W, H = (orig_img_size[0], orig_img_size[1]) if orig_img_size is not None else (is_, is_)
            lut = _keep_lut(ignore_face_idxs, dev) if ignore_face_idxs is not None else None
            flows, tile_hit = _StackedFlowFunction.apply(
                ndc, _stacked_faces(faces), cols, lut, neurenderer.fill_back, is_, neurenderer.near, neurenderer.far,
                neurenderer.rasterizer_eps, neurenderer.background_color, min(int(H), is_), min(int(W), is_),
                bool(sparse_flows), tile_out := [])
            tiles = tile_out[0]
            # the coverage bytes of the two renders ride along: a consumer that knows them (pair_consist) does not
            # even read the flows where nothing was rendered (they are exactly zero there -- or, with sparse_flows,
            # not even written: then the render's tile list rides along too)
            # (recorded with the tensor's version: an in-place write into the flows invalidates the hand-over)
            flows._hoc_coverage = (tile_hit, is_, flows._version, tiles)
            return [flows[:B], flows[B:]]
        # both renders of the pair as one launch over 2B meshes, in the training path's output set (no depth /
        # weight maps, third colour plane untouched, flow mask folded into the render)
        if USE_FLOW_RENDER and hasattr(neurenderer, "render_projected_flow") and not neurenderer.anti_aliasing:
            lut = _keep_lut(ignore_face_idxs, dev) if ignore_face_idxs is not None else None
            ro = neurenderer.render_projected_flow(ndc, _stacked_faces(faces), cols, lut)
        else:
            ro = neurenderer.render_projected_vertex_colors(ndc, _stacked_faces(faces), cols)
        return _fused_epilogue_stacked(ro, orig_img_size, ignore_face_idxs)
    # Every other setting (attached renders, materialised textures, no occlusion masking, CPU-side callers of the
    # drop-in): the same algebra from this module's own pieces, one direction at a time.
    pixels = [project.batch_proj2d(v, K) for v, K in zip(verts_cam, camintrs)]
    renders, valid = [], []
    # (only the FIRST direction's texture honours detach_textures, opticalflow.py:100-102 vs :123)
    for src, detach_tex in ((0, detach_textures), (1, False)):
        ro, m = _render_direction(neurenderer, verts_cam[src], faces, pixels[1 - src] - pixels[src], camintrs[src], detach_tex,
                                  detach_renders, ignore_face_idxs)
        renders.append(ro)
        valid.append(m)
    rgb = renders[0]["rgb"]
    if (USE_FUSED_EPILOGUE and mask_occlusions and rgb.is_cuda and rgb.dim() == 4
            and rgb.shape[2] == rgb.shape[3] == renders[0]["face_index_map"].shape[1]):
        return _fused_epilogue(renders[0], renders[1], orig_img_size, ignore_face_idxs)
    flows = [ro["rgb"] * m for ro, m in zip(renders, valid)]
    if mask_occlusions:
        # from here on the second direction's mask is the RAW alpha of its render (no threshold, no ignore list)
        # -- assigned inside the reference's no_grad block (opticalflow.py:137-139): it carries NO gradient into the render
        raw_alpha2 = renders[1]["alpha"].detach().unsqueeze(1)
        with torch.no_grad():
            visible = imgflowarp.get_occlusion_mask(valid[0], raw_alpha2, flows[0], flows[1])
        flows = [flow * (m * vis.unsqueeze(1)) for flow, m, vis in zip(flows, (valid[0], raw_alpha2), visible)]
    return [_pixels_last_xy(flow, orig_img_size) for flow in flows]


def _render_direction(neurenderer, verts, faces, displacement, camintr, detach_textures, detach_renders, ignore_face_idxs):
    """One direction of the pair: the per-vertex displacement painted as colours ``(dx, dy, 1)`` on the mesh at
    ``verts`` and rendered (opticalflow.py:100-108 / :121-126), and where that render is to be believed: opaque
    pixels whose winning face is not on the ignore list (:109-117 / :127-135)."""
    colours = torch.cat([displacement, torch.ones_like(displacement[:, :, :1])], -1)
    renderout = _render_flow(neurenderer, verts, faces, colours, camintr, detach_textures, detach_renders)
    believed = (renderout["alpha"].unsqueeze(1) > 0.99999).float()
    if ignore_face_idxs is not None:
        believed = believed * _ignore_mask(renderout["face_index_map"], ignore_face_idxs)
    return renderout, believed


def _pixels_last_xy(flow, orig_img_size):
    """[B,3,is,is] -> [B,H,W,2]: channels last, the constant third channel dropped, cropped to the frame
    (opticalflow.py:146-154; ``orig_img_size`` is (width, height))."""
    flow = flow.permute(0, 2, 3, 1)[:, :, :, :2]
    if orig_img_size is not None:
        flow = flow[:, : orig_img_size[1], : orig_img_size[0]]
    return flow
