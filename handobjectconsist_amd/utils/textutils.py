"""Per-vertex colours -> neural-renderer face textures -- stands in for
``libyana.renderutils.textutils`` (third-party, called at
/root/reference/meshreg/warping/opticalflow.py:103,123).

Layout (ASSUMED -- libyana v0.2.0 is not available to check, no golden vector of
the real helper exists): texture size 2, zero everywhere except texel (1,0,0) = colour of
vertex 0, (0,1,0) = vertex 1, (0,0,1) = vertex 2.

The SAME assumption is baked into the fused vertex-colour kernels (``mr_render_vc_*``,
``mr_render_flow_*``: csrc/raster_fwd.hip resolve step, csrc/raster_bwd.hip ``gather_vc_pixel`` and the scatter kernels), which
never materialise this tensor.  If the real libyana layout turns out to differ, set
``warping.opticalflow.USE_VERTEX_COLOR_RENDER = False`` (the flow render then goes through this
helper and the generic texture kernels, which take any [B,F,2,2,2,3] tensor) and change this
function; the vertex-colour kernels would need the matching texel -> vertex table."""
import torch


def batch_vertex_textures(faces, vertex_colors):
    """faces [B,F,3] (int), vertex_colors [B,V,3] -> textures [B,F,2,2,2,3] (differentiable)."""
    B, Fn = faces.shape[:2]
    V = vertex_colors.shape[1]
    idx = faces.long() + (torch.arange(B, device=faces.device) * V)[:, None, None]
    # index_select: its backward is an atomic index_add (no per-call index sort)
    cols = vertex_colors.reshape(B * V, 3).index_select(0, idx.reshape(-1)).view(B, Fn, 3, 3)
    tex = vertex_colors.new_zeros((B, Fn, 8, 3))
    # flat texel index = 4 * i0 + 2 * i1 + i2
    tex = torch.cat([tex[:, :, :1], cols[:, :, 2:3], cols[:, :, 1:2], tex[:, :, :1], cols[:, :, 0:1],
                     tex[:, :, :3]], dim=2)
    return tex.view(B, Fn, 2, 2, 2, 3)
