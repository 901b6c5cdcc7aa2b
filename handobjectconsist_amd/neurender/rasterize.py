"""Differentiable rasterisation on MI355X -- drop-in for meshreg/neurender/rasterize.py.

Same public names, argument order and return conventions as the reference module
(/root/reference/meshreg/neurender/rasterize.py):

* ``RasterizeFunction`` / ``Rasterize``     (rasterize.py:16, :318) -- built on the five
  upstream-compatible C-ABI entry points (``mr_forward_face_index_map`` ...), with the
  reference's buffer allocation / pre-fill / alpha / background logic (rasterize.py:60-103).
* ``rasterize_rgbad`` / ``rasterize`` / ``rasterize_silhouettes`` / ``rasterize_depth``
  (rasterize.py:362, :451, :483, :511) -- by default routed through the fused kernels
  (``mr_render_forward`` / ``mr_render_backward``): one pass writes rgb/alpha/depth already
  flipped + NCHW, the backward recomputes the sampling weights instead of storing them.
  ``USE_FUSED = False`` selects the composed reference structure instead (same results).

The native code lives in libmeshraster_hip.so (include/meshraster_hip.h); there is no CPU
or PyTorch fallback.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from handobjectconsist_amd import _lib
from handobjectconsist_amd.utils import textutils

DEFAULT_IMAGE_SIZE = 256
DEFAULT_ANTI_ALIASING = True
DEFAULT_NEAR = 0.1
DEFAULT_FAR = 100
DEFAULT_EPS = 1e-4
DEFAULT_BACKGROUND_COLOR = (0, 0, 0)

# route rasterize_rgbad through the fused kernels (True) or through RasterizeFunction (False)
USE_FUSED = True
# validation / A-B profiling: run the upstream-structured algorithms inside the fused calls
REFERENCE_ALGO = False
# profiling only (scripts/): the backward kernels' switches, `flags >> 8` of mr_render_backward
BACKWARD_DEBUG = 0


def _dummy(device, dtype=torch.float32):
    return torch.zeros(1, dtype=dtype, device=device)


class RasterizeFunction(Function):
    """The reference's autograd op (rasterize.py:16-315) on the five upstream-compatible C-ABI entry
    points: ``apply(faces, textures, image_size, near, far, eps, background_color, return_rgb,
    return_alpha, return_depth) -> (rgb [B,is,is,3], alpha [B,is,is], depth [B,is,is],
    face_index_map, face_inv_map, weight_map)``, all in RASTER orientation (un-flipped, NHWC).

    The caller-side contract of the native calls is the reference's: every output map is allocated
    and pre-filled here (index -1, depth ``far``, the rest 0; 1-element dummies for the outputs
    that were not requested -- rasterize.py:58-85), the callee only touches the pixels a face
    covers.  Background colour and alpha are applied with tensor ops afterwards (:245-260).
    """

    @staticmethod
    def forward(ctx, faces, textures, image_size, near, far, eps, background_color, return_rgb=False,
                return_alpha=False, return_depth=False):
        ctx.set_materialize_grads(False)  # unused outputs' gradients arrive as None, not as zero maps
        _lib.check_cuda(faces, textures if return_rgb else None)
        faces = _lib.contig(faces).clone()
        dev = faces.device
        B, Fn = faces.shape[:2]
        is_ = int(image_size)
        st = _lib.stream_ptr(dev)
        ctx.cfg = dict(B=B, F=Fn, is_=is_, near=float(near), far=float(far), eps=float(eps), rgb=bool(return_rgb),
                       alpha=bool(return_alpha), depth=bool(return_depth))

        def maps(shape, fill=0.0, dtype=torch.float32, wanted=True):
            return torch.full(shape, fill, dtype=dtype, device=dev) if wanted else _dummy(dev)

        textures = _lib.contig(textures) if return_rgb else _dummy(dev)
        face_index_map = maps((B, is_, is_), -1, torch.int32)
        weight_map = maps((B, is_, is_, 3))
        depth_map = maps((B, is_, is_), float(far))
        face_inv_map = maps((B, is_, is_, 3, 3), wanted=return_depth)
        rgb_map = maps((B, is_, is_, 3), wanted=return_rgb)
        sampling_index_map = maps((B, is_, is_, 8), 0, torch.int32, wanted=return_rgb)
        sampling_weight_map = maps((B, is_, is_, 8), wanted=return_rgb)
        alpha_map = maps((B, is_, is_), wanted=return_alpha)

        faces_inv = torch.zeros_like(faces)  # scratch of the native call (rasterize.py:201)
        _lib.call("mr_forward_face_index_map", _lib.ptr(faces), _lib.ptr(face_index_map), _lib.ptr(weight_map),
                  _lib.ptr(depth_map), _lib.ptr(face_inv_map), _lib.ptr(faces_inv), B, Fn, is_, float(near),
                  float(far), int(return_rgb), int(return_alpha), int(return_depth), st)
        hit = face_index_map >= 0
        if return_rgb:
            _lib.call("mr_forward_texture_sampling", _lib.ptr(faces), _lib.ptr(textures), _lib.ptr(face_index_map),
                      _lib.ptr(weight_map), _lib.ptr(depth_map), _lib.ptr(rgb_map), _lib.ptr(sampling_index_map),
                      _lib.ptr(sampling_weight_map), B, Fn, is_, int(textures.shape[2]), float(eps), st)
            bg = torch.as_tensor(background_color, dtype=torch.float32, device=dev)
            bg = bg[None, None, None, :] if bg.ndimension() == 1 else bg[:, None, None, :]
            mask = hit.float()[:, :, :, None]
            rgb_map = rgb_map * mask + (1 - mask) * bg
        if return_alpha:
            alpha_map[hit] = 1

        ctx.save_for_backward(faces, textures, face_index_map, weight_map, depth_map, rgb_map, alpha_map,
                              face_inv_map, sampling_index_map, sampling_weight_map)
        ctx.mark_non_differentiable(face_index_map)
        none = torch.tensor([])
        return (rgb_map if return_rgb else none, alpha_map.clone() if return_alpha else none,
                depth_map.clone() if return_depth else none, face_index_map, face_inv_map, weight_map)

    @staticmethod
    def backward(ctx, grad_rgb_map, grad_alpha_map, grad_depth_map, _g_index, _g_inv, _g_weight):
        (faces, textures, face_index_map, weight_map, depth_map, rgb_map, alpha_map, face_inv_map,
         sampling_index_map, sampling_weight_map) = ctx.saved_tensors
        c = ctx.cfg
        dev = faces.device
        st = _lib.stream_ptr(dev)

        def incoming(grad, like, wanted):
            # a missing gradient of a requested output counts as zeros (the reference trips over an
            # unset ctx attribute here for the depth map, rasterize.py:179 -- not reproduced)
            if not wanted:
                return _dummy(dev)
            return grad.contiguous() if grad is not None else torch.zeros_like(like)

        grad_rgb_map = incoming(grad_rgb_map, rgb_map, c["rgb"])
        grad_alpha_map = incoming(grad_alpha_map, alpha_map, c["alpha"])
        grad_depth_map = incoming(grad_depth_map, depth_map, c["depth"])
        grad_faces = torch.zeros_like(faces)
        grad_textures = torch.zeros_like(textures) if c["rgb"] else None
        if c["rgb"] or c["alpha"]:
            _lib.call("mr_backward_pixel_map", _lib.ptr(faces), _lib.ptr(face_index_map), _lib.ptr(rgb_map),
                      _lib.ptr(alpha_map), _lib.ptr(grad_rgb_map), _lib.ptr(grad_alpha_map), _lib.ptr(grad_faces),
                      c["B"], c["F"], c["is_"], c["eps"], int(c["rgb"]), int(c["alpha"]), st)
        if c["rgb"]:
            _lib.call("mr_backward_textures", _lib.ptr(face_index_map), _lib.ptr(sampling_weight_map),
                      _lib.ptr(sampling_index_map), _lib.ptr(grad_rgb_map), _lib.ptr(grad_textures), c["B"], c["F"],
                      c["is_"], int(textures.shape[2]), st)
        if c["depth"]:
            _lib.call("mr_backward_depth_map", _lib.ptr(faces), _lib.ptr(depth_map), _lib.ptr(face_index_map),
                      _lib.ptr(face_inv_map), _lib.ptr(weight_map), _lib.ptr(grad_depth_map), _lib.ptr(grad_faces),
                      c["B"], c["F"], c["is_"], st)
        if not ctx.needs_input_grad[1]:
            grad_textures = None
        return grad_faces, grad_textures, None, None, None, None, None, None, None, None


class Rasterize(nn.Module):
    """
    Wrapper around the autograd function RasterizeFunction (reference rasterize.py:318-359).
    """

    def __init__(self, image_size, near, far, eps, background_color, return_rgb=False, return_alpha=False,
                 return_depth=False):
        super(Rasterize, self).__init__()
        self.image_size = image_size
        self.near = near
        self.far = far
        self.eps = eps
        self.background_color = background_color
        self.return_rgb = return_rgb
        self.return_alpha = return_alpha
        self.return_depth = return_depth

    def forward(self, faces, textures):
        if faces.device.type == "cpu" or (textures is not None and textures.device.type == "cpu"):
            raise TypeError("Rasterize module supports only cuda Tensors")
        return RasterizeFunction.apply(faces, textures, self.image_size, self.near, self.far, self.eps,
                                       self.background_color, self.return_rgb, self.return_alpha,
                                       self.return_depth)


# ---------------------------------------------------------------------------------------
# fused path
# ---------------------------------------------------------------------------------------

_BG_CACHE = {}


def _background_tensor(background_color, device, batch_size):
    """Device copy of the background colour: [3] (stride 0) or [B,3] (stride 3)."""
    if torch.is_tensor(background_color):
        bg = background_color.to(device=device, dtype=torch.float32).contiguous()
    else:
        key = (tuple(map(float, background_color)) if not isinstance(background_color[0], (list, tuple))
               else tuple(tuple(map(float, row)) for row in background_color), str(device))
        bg = _BG_CACHE.get(key)
        if bg is None:
            bg = torch.tensor(background_color, dtype=torch.float32, device=device)
            _BG_CACHE[key] = bg
    if bg.ndimension() == 1:
        if bg.numel() != 3:
            raise ValueError("background_color must have 3 components")
        return bg, 0
    if bg.ndimension() == 2 and bg.shape == (batch_size, 3):
        return bg, 3
    raise ValueError("background_color must be [3] or [batch_size, 3]")


class RasterizeFusedFunction(Function):
    """(faces[B,F,3,3], textures[B,F,ts,ts,ts,3]) -> rgb[B,3,is,is], alpha[B,is,is],
    depth[B,is,is] in IMAGE orientation + face_index_map / weight_map in raster orientation:
    RasterizeFunction followed by the permute/flip of rasterize_rgbad, in one kernel."""

    @staticmethod
    def forward(ctx, faces, textures, image_size, near, far, eps, background_color, return_rgb, return_alpha,
                return_depth):
        ctx.set_materialize_grads(False)  # unused outputs' gradients arrive as None, not as zero maps
        _lib.check_cuda(faces, textures if return_rgb else None)
        if faces.dim() != 4 or faces.shape[2:] != (3, 3):
            raise ValueError("faces must be [batch size, number of faces, 3, 3]")
        faces = _lib.contig(faces.detach())
        dev = faces.device
        B, Fn = faces.shape[:2]
        is_ = int(image_size)
        ts = 1
        tex = None
        bg, bg_stride = None, 0
        if return_rgb:
            tex = _lib.contig(textures.detach())
            if tex.dim() != 6 or tex.shape[:2] != (B, Fn) or tex.shape[-1] != 3:
                raise ValueError("textures must be [batch size, number of faces, ts, ts, ts, 3]")
            ts = int(tex.shape[2])
            bg, bg_stride = _background_tensor(background_color, dev, B)
        empty = torch.empty
        rgb = empty((B, 3, is_, is_), dtype=torch.float32, device=dev) if return_rgb else None
        alpha = empty((B, is_, is_), dtype=torch.float32, device=dev) if return_alpha else None
        depth = empty((B, is_, is_), dtype=torch.float32, device=dev) if return_depth else None
        fim = empty((B, is_, is_), dtype=torch.int32, device=dev)
        wmap = empty((B, is_, is_, 3), dtype=torch.float32, device=dev)
        wbytes = _lib.load().mr_render_workspace_bytes(B, Fn, is_)
        work = empty((max(int(wbytes), 8),), dtype=torch.uint8, device=dev)
        flags = _lib.FLAG_REFERENCE_ALGO if REFERENCE_ALGO else 0
        _lib.call("mr_render_forward", _lib.ptr(faces), _lib.ptr(tex), _lib.ptr(bg), bg_stride, _lib.ptr(rgb),
                  _lib.ptr(alpha), _lib.ptr(depth), _lib.ptr(fim), _lib.ptr(wmap), None, _lib.ptr(work),
                  int(wbytes), B, Fn, is_, ts, float(near), float(far), float(eps), int(return_rgb),
                  int(return_alpha), int(return_depth), flags, _lib.stream_ptr(dev))
        ctx.cfg = (is_, float(near), float(far), float(eps), bool(return_rgb), bool(return_alpha),
                   bool(return_depth), ts, flags)
        ctx.save_for_backward(faces, tex if tex is not None else _dummy(dev), fim,
                              rgb if rgb is not None else _dummy(dev),
                              alpha if alpha is not None else _dummy(dev))
        ctx.mark_non_differentiable(fim, wmap)
        e = torch.tensor([])
        return (rgb if return_rgb else e, alpha if return_alpha else e, depth if return_depth else e, fim,
                wmap)

    @staticmethod
    def backward(ctx, grad_rgb, grad_alpha, grad_depth, _gfim, _gw):
        faces, tex, fim, rgb, alpha = ctx.saved_tensors
        is_, near, far, eps, rr, ra, rd, ts, flags = ctx.cfg
        flags |= BACKWARD_DEBUG << 8
        dev = faces.device
        B, Fn = faces.shape[:2]
        want_faces, want_tex = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and rr
        grad_faces = grad_textures = None
        if not (want_faces or want_tex):
            return (None,) * 10
        g_rgb = _lib.contig(grad_rgb) if (rr and grad_rgb is not None) else None
        g_alpha = _lib.contig(grad_alpha) if (ra and grad_alpha is not None) else None
        g_depth = _lib.contig(grad_depth) if (rd and grad_depth is not None) else None
        if want_faces:
            grad_faces = torch.empty_like(faces)
        if want_tex:
            if g_rgb is None:
                grad_textures = torch.zeros_like(tex)
                want_tex = False
            else:
                grad_textures = torch.empty_like(tex)
        if want_faces or want_tex:
            if want_faces and (g_rgb is not None or g_alpha is not None):  # kernel D runs: owner flags / list / records
                wbytes = int(_lib.load().mr_render_backward_workspace_bytes(B, Fn, is_))
            else:  # the list of the faces that own a pixel (the gather walks only those)
                wbytes = int(_lib.load().mr_render_backward_list_workspace_bytes(B, Fn))
            work = torch.empty((max(wbytes, 8),), dtype=torch.uint8, device=dev)
            _lib.call("mr_render_backward", _lib.ptr(faces), _lib.ptr(tex) if rr else None, _lib.ptr(fim),
                      _lib.ptr(rgb) if rr else None, _lib.ptr(alpha) if ra else None, _lib.ptr(g_rgb),
                      _lib.ptr(g_alpha), _lib.ptr(g_depth), _lib.ptr(grad_faces),
                      _lib.ptr(grad_textures) if want_tex else None, _lib.ptr(work), wbytes, B, Fn, is_, ts, near, far, eps,
                      int(rr), int(ra), int(rd), flags, _lib.stream_ptr(dev))
        return grad_faces, grad_textures, None, None, None, None, None, None, None, None


class RasterizeVertexColorFunction(Function):
    """Render of per-vertex colours straight from (projected vertices, vertex indices, colours):
    = batch_vertex_textures -> fill-back -> vertices_to_faces -> RasterizeFusedFunction, without
    materialising the face coordinates, the 2x2x2 textures or their fill-back copies
    (mr_render_vc_forward / mr_render_vc_backward).  Differentiable w.r.t. the colours only --
    the vertex positions must be detached (the reference's training setting, detach_renders=True,
    warpbranch.py:65-66)."""

    @staticmethod
    def forward(ctx, verts_ndc, faces_idx, vcolors, fill_back, image_size, near, far, eps, background_color,
                return_rgb, return_alpha, return_depth):
        ctx.set_materialize_grads(False)  # unused outputs' gradients arrive as None, not as zero maps
        _lib.check_cuda(verts_ndc, faces_idx, vcolors)
        if not (float(eps) >= 1e-6):
            raise ValueError("vertex-colour rendering needs eps >= 1e-6")
        verts = _lib.contig(verts_ndc.detach())
        fidx = faces_idx.detach().to(torch.int32).contiguous()
        cols = _lib.contig(vcolors.detach())
        dev = verts.device
        B, V = verts.shape[:2]
        F0 = fidx.shape[1]
        if fidx.shape != (B, F0, 3) or cols.shape != (B, V, 3) or verts.shape != (B, V, 3):
            raise ValueError("expected vertices [B,V,3], faces [B,F,3], vertex colours [B,V,3]")
        is_ = int(image_size)
        bg, bg_stride = _background_tensor(background_color, dev, B) if return_rgb else (None, 0)
        empty = torch.empty
        rgb = empty((B, 3, is_, is_), dtype=torch.float32, device=dev) if return_rgb else None
        alpha = empty((B, is_, is_), dtype=torch.float32, device=dev) if return_alpha else None
        depth = empty((B, is_, is_), dtype=torch.float32, device=dev) if return_depth else None
        fim = empty((B, is_, is_), dtype=torch.int32, device=dev)
        wmap = empty((B, is_, is_, 3), dtype=torch.float32, device=dev)
        F = 2 * F0 if fill_back else F0
        wbytes = _lib.load().mr_render_workspace_bytes(B, F, is_)
        work = empty((max(int(wbytes), 8),), dtype=torch.uint8, device=dev)
        _lib.call("mr_render_vc_forward", _lib.ptr(verts), _lib.ptr(fidx), _lib.ptr(cols), _lib.ptr(bg), bg_stride,
                  _lib.ptr(rgb), _lib.ptr(alpha), _lib.ptr(depth), _lib.ptr(fim), _lib.ptr(wmap), _lib.ptr(work),
                  int(wbytes), B, V, F0, int(bool(fill_back)), is_, float(near), float(far), float(eps),
                  int(return_rgb), int(return_alpha), int(return_depth), 0, textutils.texel_layout_code(), _lib.stream_ptr(dev))
        ctx.cfg = (is_, float(eps), bool(fill_back), bool(return_rgb), bool(return_depth))
        # the forward's own weight / depth maps feed the backward (no extra memory: they are outputs)
        ctx.save_for_backward(verts, fidx, fim, wmap, depth if return_depth else fim)
        ctx.mark_non_differentiable(fim, wmap)
        e = torch.tensor([])
        return (rgb if return_rgb else e, alpha if return_alpha else e, depth if return_depth else e, fim, wmap)

    @staticmethod
    def backward(ctx, grad_rgb, _ga, _gd, _gf, _gw):
        verts, fidx, fim, wmap, depth = ctx.saved_tensors
        is_, eps, fill_back, rr, rd = ctx.cfg
        if not ctx.needs_input_grad[2] or not rr:
            return (None,) * 12
        B, V = verts.shape[:2]
        grad_cols = torch.empty((B, V, 3), dtype=torch.float32, device=verts.device)
        if grad_rgb is None:
            grad_cols.zero_()
        else:
            g = _lib.contig(grad_rgb)
            _lib.call("mr_render_vc_backward", _lib.ptr(verts), _lib.ptr(fidx), _lib.ptr(fim),
                      _lib.ptr(wmap) if rd else None, _lib.ptr(depth) if rd else None, _lib.ptr(g), _lib.ptr(grad_cols), B, V, int(fidx.shape[1]), int(fill_back), is_, eps, 0, textutils.texel_layout_code(),
                      _lib.stream_ptr(verts.device))
        return (None, None, grad_cols) + (None,) * 9


class RasterizeFlowFunction(Function):
    """``RasterizeVertexColorFunction`` restricted to what ``get_opticalflow`` consumes from a render
    (mr_render_flow_forward): rgb [B,3,is,is] whose first two planes hold the rendered displacement (the third
    is never written), alpha, the flow mask ``(alpha > 0.99999) * keep_lut[face + 1]`` of opticalflow.py:109-117
    and face_index_map.  The depth image and the weight map are written at covered pixels only (the backward w.r.t.
    the colours reads them there and nowhere else; they are not returned): 20 instead of 36 bytes per background
    pixel leave the kernel, and the separate mask kernel is gone."""

    @staticmethod
    def forward(ctx, verts_ndc, faces_idx, vcolors, keep_lut, fill_back, image_size, near, far, eps, background_color,
                alpha_thresh):
        ctx.set_materialize_grads(False)
        _lib.check_cuda(verts_ndc, faces_idx, vcolors, keep_lut)
        if not (float(eps) >= 1e-6):
            raise ValueError("vertex-colour rendering needs eps >= 1e-6")
        verts = _lib.contig(verts_ndc.detach())
        fidx = faces_idx.detach().to(torch.int32).contiguous()
        cols = _lib.contig(vcolors.detach())
        dev = verts.device
        B, V = verts.shape[:2]
        F0 = fidx.shape[1]
        if fidx.shape != (B, F0, 3) or cols.shape != (B, V, 3) or verts.shape != (B, V, 3):
            raise ValueError("expected vertices [B,V,3], faces [B,F,3], vertex colours [B,V,3]")
        is_ = int(image_size)
        bg, bg_stride = _background_tensor(background_color, dev, B)
        lut = _lib.contig(keep_lut) if keep_lut is not None else None
        empty = torch.empty
        rgb = empty((B, 3, is_, is_), dtype=torch.float32, device=dev)
        alpha = empty((B, is_, is_), dtype=torch.float32, device=dev)
        mask = empty((B, is_, is_), dtype=torch.float32, device=dev)
        fim = empty((B, is_, is_), dtype=torch.int32, device=dev)
        depth = empty((B, is_, is_), dtype=torch.float32, device=dev)  # valid where fim >= 0 only
        wmap = empty((B, is_, is_, 3), dtype=torch.float32, device=dev)  # valid where fim >= 0 only
        F = 2 * F0 if fill_back else F0
        wbytes = _lib.load().mr_render_workspace_bytes(B, F, is_)
        work = empty((max(int(wbytes), 8),), dtype=torch.uint8, device=dev)
        _lib.call("mr_render_flow_forward", _lib.ptr(verts), _lib.ptr(fidx), _lib.ptr(cols), _lib.ptr(bg), bg_stride,
                  _lib.ptr(lut), int(lut.numel()) if lut is not None else 0, float(alpha_thresh), _lib.ptr(rgb),
                  _lib.ptr(alpha), _lib.ptr(mask), _lib.ptr(depth), _lib.ptr(wmap), _lib.ptr(fim), None, _lib.ptr(work), int(wbytes),
                  B, V, F0, int(bool(fill_back)), is_, float(near), float(far), float(eps), 0, None, 0, None, None, 0, textutils.texel_layout_code(), _lib.stream_ptr(dev))
        ctx.cfg = (is_, float(eps), bool(fill_back))
        ctx.save_for_backward(verts, fidx, fim, wmap, depth)
        ctx.mark_non_differentiable(alpha, mask, fim)
        return rgb, alpha, mask, fim

    @staticmethod
    def backward(ctx, grad_rgb, _ga, _gm, _gf):
        verts, fidx, fim, wmap, depth = ctx.saved_tensors
        is_, eps, fill_back = ctx.cfg
        if not ctx.needs_input_grad[2]:
            return (None,) * 11
        B, V = verts.shape[:2]
        grad_cols = torch.empty((B, V, 3), dtype=torch.float32, device=verts.device)
        if grad_rgb is None:
            grad_cols.zero_()
        else:
            g = _lib.contig(grad_rgb)
            _lib.call("mr_render_vc_backward", _lib.ptr(verts), _lib.ptr(fidx), _lib.ptr(fim), _lib.ptr(wmap), _lib.ptr(depth),
                      _lib.ptr(g), _lib.ptr(grad_cols), B, V, int(fidx.shape[1]), int(fill_back), is_, eps, 0, textutils.texel_layout_code(),
                      _lib.stream_ptr(verts.device))
        return (None, None, grad_cols) + (None,) * 8


def rasterize_flow(vertices_ndc, faces_idx, vertex_colors, keep_lut=None, fill_back=True,
                   image_size=DEFAULT_IMAGE_SIZE, near=DEFAULT_NEAR, far=DEFAULT_FAR, eps=DEFAULT_EPS,
                   background_color=DEFAULT_BACKGROUND_COLOR, alpha_thresh=0.99999):
    """The training-path render of get_opticalflow: {'rgb' (planes 0 and 1 valid), 'alpha', 'mask',
    'face_index_map'} -- see ``RasterizeFlowFunction``.  No anti-aliasing."""
    if background_color is None:
        background_color = DEFAULT_BACKGROUND_COLOR
    rgb, alpha, mask, fim = RasterizeFlowFunction.apply(vertices_ndc, faces_idx, vertex_colors, keep_lut, fill_back,
                                                        image_size, near, far, eps, background_color, alpha_thresh)
    return {"rgb": rgb, "alpha": alpha, "mask": mask, "face_index_map": fim}


def rasterize_vertex_colors(
    vertices_ndc,
    faces_idx,
    vertex_colors,
    fill_back=True,
    image_size=DEFAULT_IMAGE_SIZE,
    anti_aliasing=DEFAULT_ANTI_ALIASING,
    near=DEFAULT_NEAR,
    far=DEFAULT_FAR,
    eps=DEFAULT_EPS,
    background_color=DEFAULT_BACKGROUND_COLOR,
):
    """rasterize_rgbad for vertex-colour textures: same returned dict as
    ``rasterize_rgbad(vertices_to_faces(v, fill_back(faces)), fill_back(batch_vertex_textures(faces,
    colours)), ...)`` (bit-identical images and maps), computed by the fused vertex-colour kernels."""
    ras_size = image_size * 2 if anti_aliasing else image_size
    if background_color is None:
        background_color = DEFAULT_BACKGROUND_COLOR
    rgb, alpha, depth, face_index_map, weight_map = RasterizeVertexColorFunction.apply(
        vertices_ndc, faces_idx, vertex_colors, fill_back, ras_size, near, far, eps, background_color, True, True, True)
    if anti_aliasing:
        rgb = F.avg_pool2d(rgb, kernel_size=(2, 2))
        alpha = F.avg_pool2d(alpha[:, None, :, :], kernel_size=(2, 2))[:, 0]
        depth = F.avg_pool2d(depth[:, None, :, :], kernel_size=(2, 2))[:, 0]
    ret = _RenderOutput({"rgb": rgb, "alpha": alpha, "depth": depth, "face_inv_map": None,
                         "face_index_map": face_index_map, "weight_map": weight_map})
    v_d, f_d = vertices_ndc.detach(), faces_idx.detach()

    def thunk():
        from handobjectconsist_amd.neurender import nr_ops

        f_all = torch.cat((f_d, f_d.flip(-1)), dim=1) if fill_back else f_d
        return face_inv_map_from(nr_ops.vertices_to_faces(v_d, f_all), face_index_map)

    ret._thunk = thunk
    return ret


class _RenderOutput(dict):
    """The dict rasterize_rgbad returns.  ``face_inv_map`` ([B,is,is,3,3], 36 B/pixel, only
    ever consumed by the depth backward, which recomputes it) is materialised on first
    access instead of being written by every forward pass."""

    _LAZY = "face_inv_map"
    _thunk = None

    def _materialise(self):
        if self._thunk is not None:
            dict.__setitem__(self, self._LAZY, self._thunk())
            self._thunk = None

    def __getitem__(self, key):
        if key == self._LAZY:
            self._materialise()
        return dict.__getitem__(self, key)

    def get(self, key, default=None):
        if key == self._LAZY:
            self._materialise()
        return dict.get(self, key, default)

    def items(self):
        self._materialise()
        return dict.items(self)

    def values(self):
        self._materialise()
        return dict.values(self)


def face_inv_map_from(faces, face_index_map):
    """[B,is,is,3,3] per-pixel inverse of the winning face (zeros on background)."""
    faces = _lib.contig(faces.detach())
    B, Fn = faces.shape[:2]
    is_ = face_index_map.shape[1]
    out = torch.empty((B, is_, is_, 3, 3), dtype=torch.float32, device=faces.device)
    _lib.call("mr_face_inv_map", _lib.ptr(faces), _lib.ptr(face_index_map), _lib.ptr(out), B, Fn, is_,
              _lib.stream_ptr(faces.device))
    return out


def rasterize_rgbad(
    faces,
    textures=None,
    image_size=DEFAULT_IMAGE_SIZE,
    anti_aliasing=DEFAULT_ANTI_ALIASING,
    near=DEFAULT_NEAR,
    far=DEFAULT_FAR,
    eps=DEFAULT_EPS,
    background_color=DEFAULT_BACKGROUND_COLOR,
    return_rgb=True,
    return_alpha=True,
    return_depth=True,
):
    """
    Generate RGB, alpha channel, and depth images from faces and textures (for RGB).
    Same contract as the reference (rasterize.py:362-448).

    Args:
        faces (torch.Tensor): [batch size, number of faces, 3 (vertices), 3 (XYZ)].
        textures (torch.Tensor): [batch size, number of faces, ts, ts, ts, 3 (RGB)].
        image_size (int): Width and height of rendered images.
        anti_aliasing (bool): do anti-aliasing by 2x super-sampling.
        near / far (float): z-range to draw.
        eps (float): small epsilon for approximated differentiation.
        background_color (tuple): background color of RGB images.
        return_rgb / return_alpha / return_depth (bool): which images to generate.

    Returns:
        dict: 'rgb' [B,3,is,is], 'alpha' [B,is,is], 'depth' [B,is,is] (image orientation),
        'face_index_map', 'weight_map', 'face_inv_map' (raster orientation, un-flipped).
    """
    if faces.device.type == "cpu" or (textures is not None and textures.device.type == "cpu"):
        raise TypeError("Rasterize module supports only cuda Tensors")
    ras_size = image_size * 2 if anti_aliasing else image_size
    if background_color is None:
        background_color = DEFAULT_BACKGROUND_COLOR

    if USE_FUSED:
        rgb, alpha, depth, face_index_map, weight_map = RasterizeFusedFunction.apply(
            faces, textures, ras_size, near, far, eps, background_color, return_rgb, return_alpha, return_depth)
        face_inv_map = None
    else:
        rgb, alpha, depth, face_index_map, face_inv_map, weight_map = Rasterize(
            ras_size, near, far, eps, background_color, return_rgb, return_alpha, return_depth)(faces, textures)
        # transpose & vertical flip (rasterize.py:413-428)
        if return_rgb:
            rgb = rgb.permute((0, 3, 1, 2)).flip(2)
        if return_alpha:
            alpha = alpha.flip(1)
        if return_depth:
            depth = depth.flip(1)

    if anti_aliasing:
        # 0.5x down-sampling
        if return_rgb:
            rgb = F.avg_pool2d(rgb, kernel_size=(2, 2))
        if return_alpha:
            alpha = F.avg_pool2d(alpha[:, None, :, :], kernel_size=(2, 2))[:, 0]
        if return_depth:
            depth = F.avg_pool2d(depth[:, None, :, :], kernel_size=(2, 2))[:, 0]

    ret = _RenderOutput({
        "rgb": rgb if return_rgb else None,
        "alpha": alpha if return_alpha else None,
        "depth": depth if return_depth else None,
        "face_inv_map": face_inv_map,
        "face_index_map": face_index_map,
        "weight_map": weight_map,
    })
    if USE_FUSED:
        if return_depth:
            faces_d, fim_d = faces.detach(), face_index_map
            ret._thunk = lambda: face_inv_map_from(faces_d, fim_d)
        else:
            dict.__setitem__(ret, "face_inv_map", _dummy(faces.device))
    return ret


def rasterize(
    faces,
    textures,
    image_size=DEFAULT_IMAGE_SIZE,
    anti_aliasing=DEFAULT_ANTI_ALIASING,
    near=DEFAULT_NEAR,
    far=DEFAULT_FAR,
    eps=DEFAULT_EPS,
    background_color=DEFAULT_BACKGROUND_COLOR,
):
    """Generate RGB images from faces and textures: [batch size, 3, image_size, image_size]."""
    return rasterize_rgbad(
        faces, textures, image_size, anti_aliasing, near, far, eps, background_color, True, False, False
    )["rgb"]


def rasterize_silhouettes(
    faces,
    image_size=DEFAULT_IMAGE_SIZE,
    anti_aliasing=DEFAULT_ANTI_ALIASING,
    near=DEFAULT_NEAR,
    far=DEFAULT_FAR,
    eps=DEFAULT_EPS,
):
    """Generate alpha channels from faces: [batch size, image_size, image_size]."""
    return rasterize_rgbad(faces, None, image_size, anti_aliasing, near, far, eps, None, False, True, False)[
        "alpha"
    ]


def rasterize_depth(
    faces,
    image_size=DEFAULT_IMAGE_SIZE,
    anti_aliasing=DEFAULT_ANTI_ALIASING,
    near=DEFAULT_NEAR,
    far=DEFAULT_FAR,
    eps=DEFAULT_EPS,
):
    """Generate depth images from faces: [batch size, image_size, image_size]."""
    return rasterize_rgbad(faces, None, image_size, anti_aliasing, near, far, eps, None, False, False, True)[
        "depth"
    ]
