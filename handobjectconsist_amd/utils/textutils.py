"""Per-vertex colours -> neural-renderer face textures -- stands in for
``libyana.renderutils.textutils`` (third-party, called at
/root/reference/meshreg/warping/opticalflow.py:103,123).

Layout (ASSUMED -- libyana v0.2.0 is not available to check, no golden vector of
the real helper exists): texture size 2, zero everywhere except three texels holding the
colours of the face's three vertices.  WHICH texel holds which vertex is data, not code:
``TEXEL_VERTEX = (s0, s1, s2)`` says texel (1,0,0) = colour of vertex s0, (0,1,0) = vertex s1,
(0,0,1) = vertex s2; the assumed layout is the identity (0, 1, 2).

This module's ``batch_vertex_textures`` (the materialised tensor of the generic texture path) and
the fused vertex-colour kernels (``mr_render_vc_*``, ``mr_render_flow_*``, which never materialise
it and take the table as their ``texel_layout`` argument, include/meshraster_hip.h) read the SAME
table: if the real libyana layout turns out to differ, change ``TEXEL_VERTEX`` -- no kernel edit.
tests/test_gpu_raster.py::test_vertex_colour_kernels_follow_the_texel_table checks the fused kernels
against the generic path for every permutation."""
import torch

TEXEL_VERTEX = (0, 1, 2)


def texel_layout_code(table=None):
    """The C-ABI's ``texel_layout`` argument for ``table`` (default: ``TEXEL_VERTEX``): two bits per texel axis."""
    s0, s1, s2 = TEXEL_VERTEX if table is None else table
    if sorted((s0, s1, s2)) != [0, 1, 2]:
        raise ValueError("the texel table must be a permutation of (0, 1, 2)")
    return int(s0) | int(s1) << 2 | int(s2) << 4


def batch_vertex_textures(faces, vertex_colors, table=None):
    """faces [B,F,3] (int), vertex_colors [B,V,3] -> textures [B,F,2,2,2,3] (differentiable)."""
    s0, s1, s2 = TEXEL_VERTEX if table is None else table
    B, Fn = faces.shape[:2]
    V = vertex_colors.shape[1]
    idx = faces.long() + (torch.arange(B, device=faces.device) * V)[:, None, None]
    # index_select: its backward is an atomic index_add (no per-call index sort)
    cols = vertex_colors.reshape(B * V, 3).index_select(0, idx.reshape(-1)).view(B, Fn, 3, 3)
    zero = vertex_colors.new_zeros((B, Fn, 3, 3))
    # flat texel index = 4 * i0 + 2 * i1 + i2: texel (0,0,1) is slot 1, (0,1,0) slot 2, (1,0,0) slot 4
    tex = torch.cat([zero[:, :, :1], cols[:, :, s2:s2 + 1], cols[:, :, s1:s1 + 1], zero[:, :, :1], cols[:, :, s0:s0 + 1],
                     zero[:, :, :3]], dim=2)
    return tex.view(B, Fn, 2, 2, 2, 3)
