"""Load the REFERENCE's own render / flow / warp glue on CPU, with the absent third-party modules
stubbed by the oracle.

TEST INFRASTRUCTURE ONLY, and BUILD-CONTAINER ONLY: needs /root/reference, which does not exist on
the GPU box.  Used by tests/golden/make_golden_chain.py to produce fixtures; nothing in the shipped
package, in the ``-m gpu`` tests, in smoke() or in bench.py imports it.

What runs unmodified from /root/reference after ``install()``:
    meshreg/neurender/rasterize.py   RasterizeFunction (forward :23-125, backward :127-197),
                                     Rasterize, rasterize_rgbad / rasterize / _silhouettes / _depth
    meshreg/neurender/renderer.py    Renderer (render, render_rgb, render_silhouettes, render_depth, project)
    meshreg/warping/opticalflow.py   get_opticalflow(s) incl. the mask algebra / flip / crop quirks
    meshreg/warping/imgflowarp.py    warp, pair_consist, get_occlusion_mask  (pure torch, no stub)
    meshreg/optim/pyramidloss.py     PyramidCriterion("l1")                  (kornia ctor symbols stubbed)
    meshreg/models/warpbranch.py     forward (GT-ref substitution, detach of frames > 0, stack().mean())

What is stubbed (the source is NOT under /root/reference; PARITY UNPINNED for these, as in
raster_oracle.c / raster_ref.py):
    neural_renderer.cuda.rasterize   the five entry points -> oracle/raster_oracle.c through ctypes, on
                                     CPU tensors, caller-allocates / callee-mutates exactly as the
                                     call sites rasterize.py:202-315 expect
    neural_renderer (python helpers) projection, vertices_to_faces, lighting, look_at, look, perspective
                                     -> differentiable torch restatements of SURVEY appendix B.1-B.3
    libyana                          camutils.project.batch_proj2d, renderutils.textutils.batch_vertex_textures
                                     (texel layout ASSUMED), renderutils.catmesh.batch_cat_meshes
    kornia                           only the two symbols pyramidloss.py:3-4 imports (never reached by "l1")
    torch.cuda.FloatTensor / IntTensor -> torch.FloatTensor / IntTensor;  Tensor.cuda() -> identity
"""
import ctypes
import math
import sys
import types

import torch

from oracle import raster_ref

REFERENCE_ROOT = "/root/reference"


# ---------------------------------------------------------------------------------------------------
# neural_renderer.cuda.rasterize: the five native entry points on CPU tensors
# ---------------------------------------------------------------------------------------------------


def _p(t):
    assert t.device.type == "cpu" and t.is_contiguous(), "stub expects contiguous CPU tensors (CHECK_CONTIGUOUS upstream)"
    return ctypes.c_void_p(t.data_ptr())


def _forward_face_index_map(faces, face_index_map, weight_map, depth_map, face_inv_map, faces_inv, image_size,
                            near, far, return_rgb, return_alpha, return_depth):
    B, F = faces.shape[:2]
    raster_ref.lib().oracle_forward_face_index_map(
        _p(faces), _p(face_index_map), _p(weight_map), _p(depth_map), _p(face_inv_map), _p(faces_inv),
        ctypes.c_int(B), ctypes.c_int(F), ctypes.c_int(int(image_size)), ctypes.c_float(near), ctypes.c_float(far),
        ctypes.c_int(int(return_rgb)), ctypes.c_int(int(return_alpha)), ctypes.c_int(int(return_depth)),
        ctypes.c_int(1))
    return face_index_map, weight_map, depth_map, face_inv_map


def _forward_texture_sampling(faces, textures, face_index_map, weight_map, depth_map, rgb_map, sampling_index_map,
                              sampling_weight_map, image_size, eps):
    B, F = faces.shape[:2]
    raster_ref.lib().oracle_forward_texture_sampling(
        _p(faces), _p(textures), _p(face_index_map), _p(weight_map), _p(depth_map), _p(rgb_map),
        _p(sampling_index_map), _p(sampling_weight_map), ctypes.c_int(B), ctypes.c_int(F),
        ctypes.c_int(int(image_size)), ctypes.c_int(textures.shape[2]), ctypes.c_float(eps))
    return rgb_map, sampling_index_map, sampling_weight_map


def _backward_pixel_map(faces, face_index_map, rgb_map, alpha_map, grad_rgb_map, grad_alpha_map, grad_faces,
                        image_size, eps, return_rgb, return_alpha):
    B, F = faces.shape[:2]
    raster_ref.lib().oracle_backward_pixel_map(
        _p(faces), _p(face_index_map), _p(rgb_map), _p(alpha_map), _p(grad_rgb_map), _p(grad_alpha_map),
        _p(grad_faces), ctypes.c_int(B), ctypes.c_int(F), ctypes.c_int(int(image_size)), ctypes.c_float(eps),
        ctypes.c_int(int(return_rgb)), ctypes.c_int(int(return_alpha)), ctypes.c_int(1))
    return grad_faces


def _backward_textures(face_index_map, sampling_weight_map, sampling_index_map, grad_rgb_map, grad_textures,
                       num_faces):
    B, is_ = face_index_map.shape[:2]
    raster_ref.lib().oracle_backward_textures(
        _p(face_index_map), _p(sampling_weight_map), _p(sampling_index_map), _p(grad_rgb_map), _p(grad_textures),
        ctypes.c_int(B), ctypes.c_int(int(num_faces)), ctypes.c_int(is_), ctypes.c_int(grad_textures.shape[2]))
    return grad_textures


def _backward_depth_map(faces, depth_map, face_index_map, face_inv_map, weight_map, grad_depth_map, grad_faces,
                        image_size):
    B, F = faces.shape[:2]
    raster_ref.lib().oracle_backward_depth_map(
        _p(faces), _p(depth_map), _p(face_index_map), _p(face_inv_map), _p(weight_map), _p(grad_depth_map),
        _p(grad_faces), ctypes.c_int(B), ctypes.c_int(F), ctypes.c_int(int(image_size)))
    return grad_faces


# ---------------------------------------------------------------------------------------------------
# neural_renderer python helpers, differentiable
# ---------------------------------------------------------------------------------------------------


def nr_projection(vertices, K, R, t, dist_coeffs, orig_size, eps=1e-9):
    """B.1.  Called at renderer.py:187 (and :146)."""
    vertices = torch.matmul(vertices, R.transpose(2, 1)) + t
    x, y, z = vertices[:, :, 0], vertices[:, :, 1], vertices[:, :, 2]
    x_ = x / (z + eps)
    y_ = y / (z + eps)
    k1, k2, p1, p2, k3 = (dist_coeffs[:, None, i] for i in range(5))
    r = torch.sqrt(x_ ** 2 + y_ ** 2)
    x__ = x_ * (1 + k1 * (r ** 2) + k2 * (r ** 4) + k3 * (r ** 6)) + 2 * p1 * x_ * y_ + p2 * (r ** 2 + 2 * x_ ** 2)
    y__ = y_ * (1 + k1 * (r ** 2) + k2 * (r ** 4) + k3 * (r ** 6)) + p1 * (r ** 2 + 2 * y_ ** 2) + 2 * p2 * x_ * y_
    vertices = torch.stack([x__, y__, torch.ones_like(z)], dim=-1)
    vertices = torch.matmul(vertices, K.transpose(1, 2))
    u, v = vertices[:, :, 0], vertices[:, :, 1]
    v = orig_size - v
    u = 2 * (u - orig_size / 2.0) / orig_size
    v = 2 * (v - orig_size / 2.0) / orig_size
    return torch.stack([u, v, z], dim=-1)


def nr_vertices_to_faces(vertices, faces):
    """B.2: vertices[b, faces[b, f, k]] -> [B, F, 3, 3]."""
    bs, nv = vertices.shape[:2]
    offs = (torch.arange(bs, dtype=torch.int64) * nv)[:, None, None]
    return vertices.reshape(bs * nv, 3)[faces.long() + offs]


def nr_lighting(faces, textures, intensity_ambient=0.5, intensity_directional=0.5, color_ambient=(1, 1, 1),
                color_directional=(1, 1, 1), direction=(0, 1, 0)):
    """B.3: light = ia*ca + id*cd*relu(n.dir), n = normalize(cross(v0-v1, v2-v1), eps=1e-5); textures * light."""
    bs, nf = faces.shape[:2]
    color_ambient = torch.as_tensor(color_ambient, dtype=torch.float32)
    color_directional = torch.as_tensor(color_directional, dtype=torch.float32)
    direction = torch.as_tensor(direction, dtype=torch.float32)
    if color_ambient.dim() == 1:
        color_ambient = color_ambient[None, :]
    if color_directional.dim() == 1:
        color_directional = color_directional[None, :]
    if direction.dim() == 1:
        direction = direction[None, :]
    light = torch.zeros(bs, nf, 3, dtype=torch.float32)
    if intensity_ambient != 0:
        light = light + intensity_ambient * color_ambient[:, None, :]
    if intensity_directional != 0:
        f = faces.reshape(bs * nf, 3, 3)
        v10 = f[:, 0] - f[:, 1]
        v12 = f[:, 2] - f[:, 1]
        normals = torch.nn.functional.normalize(torch.cross(v10, v12, dim=1), eps=1e-5).reshape(bs, nf, 3)
        cos = torch.nn.functional.relu(torch.sum(normals * direction[:, None, :], dim=2))
        light = light + intensity_directional * (color_directional[:, None, :] * cos[:, :, None])
    return textures * light[:, :, None, None, None, :]


def nr_look_at(vertices, eye, at=(0, 0, 0), up=(0, 1, 0)):
    eye = torch.as_tensor(eye, dtype=torch.float32)[None, :]
    at = torch.as_tensor(at, dtype=torch.float32)[None, :]
    up = torch.as_tensor(up, dtype=torch.float32)[None, :]
    z_axis = torch.nn.functional.normalize(at - eye, eps=1e-5)
    x_axis = torch.nn.functional.normalize(torch.cross(up, z_axis, dim=1), eps=1e-5)
    y_axis = torch.nn.functional.normalize(torch.cross(z_axis, x_axis, dim=1), eps=1e-5)
    r = torch.cat((x_axis[:, None, :], y_axis[:, None, :], z_axis[:, None, :]), dim=1)
    return torch.matmul(vertices - eye[:, None, :], r.transpose(1, 2))


def nr_look(vertices, eye, direction=(0, 1, 0), up=None):
    direction = torch.as_tensor(direction, dtype=torch.float32)[None, :]
    eye = torch.as_tensor(eye, dtype=torch.float32)[None, :]
    up = torch.as_tensor((0, 1, 0) if up is None else up, dtype=torch.float32)[None, :]
    z_axis = torch.nn.functional.normalize(direction, eps=1e-5)
    x_axis = torch.nn.functional.normalize(torch.cross(up, z_axis, dim=1), eps=1e-5)
    y_axis = torch.nn.functional.normalize(torch.cross(z_axis, x_axis, dim=1), eps=1e-5)
    r = torch.cat((x_axis[:, None, :], y_axis[:, None, :], z_axis[:, None, :]), dim=1)
    return torch.matmul(vertices - eye[:, None, :], r.transpose(1, 2))


def nr_perspective(vertices, angle=30.0):
    width = math.tan(math.radians(angle))
    z = vertices[:, :, 2]
    return torch.stack((vertices[:, :, 0] / z / width, vertices[:, :, 1] / z / width, z), dim=2)


# ---------------------------------------------------------------------------------------------------
# libyana helpers on the path
# ---------------------------------------------------------------------------------------------------


def batch_proj2d(verts, camintr, camextr=None):
    """libyana.camutils.project.batch_proj2d (opticalflow.py:98-99): h = K v^T; h[:, :, :2] / h[:, :, 2:]."""
    if camextr is not None:
        verts = camextr[:, :3, :3].bmm(verts.transpose(1, 2)).transpose(1, 2) + camextr[:, :3, 3].unsqueeze(1)
    hom = camintr.bmm(verts.transpose(1, 2)).transpose(1, 2)
    return hom[:, :, :2] / hom[:, :, 2:]


def batch_vertex_textures(faces, vertex_colors):
    """libyana.renderutils.textutils.batch_vertex_textures (opticalflow.py:103,123) -> [B,F,2,2,2,3] with the
    three vertex colours at texels (1,0,0), (0,1,0), (0,0,1) -- layout ASSUMED (source absent)."""
    B, F = faces.shape[:2]
    idx = faces.long().reshape(B, F * 3, 1).expand(-1, -1, vertex_colors.shape[-1])
    fc = torch.gather(vertex_colors, 1, idx).reshape(B, F, 3, vertex_colors.shape[-1])
    tex = vertex_colors.new_zeros(B, F, 2, 2, 2, vertex_colors.shape[-1])
    tex[:, :, 1, 0, 0] = fc[:, :, 0]
    tex[:, :, 0, 1, 0] = fc[:, :, 1]
    tex[:, :, 0, 0, 1] = fc[:, :, 2]
    return tex


def batch_cat_meshes(verts_list, faces_list):
    """libyana.renderutils.catmesh.batch_cat_meshes (warpbranch.py:50)."""
    off, faces = 0, []
    for v, f in zip(verts_list, faces_list):
        faces.append(f + off)
        off += v.shape[1]
    return torch.cat(verts_list, 1), torch.cat(faces, 1), None


# ---------------------------------------------------------------------------------------------------


def _module(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


_INSTALLED = False


def install():
    """Put the stubs into sys.modules, patch torch for CPU execution and make /root/reference importable.
    Returns a namespace with the reference's modules."""
    global _INSTALLED
    sys.dont_write_bytecode = True  # the reference tree is read-only
    if not _INSTALLED:
        if REFERENCE_ROOT not in sys.path:
            sys.path.insert(0, REFERENCE_ROOT)
        rast = _module("neural_renderer.cuda.rasterize", forward_face_index_map=_forward_face_index_map,
                       forward_texture_sampling=_forward_texture_sampling, backward_pixel_map=_backward_pixel_map,
                       backward_textures=_backward_textures, backward_depth_map=_backward_depth_map)
        cuda = _module("neural_renderer.cuda", rasterize=rast)
        _module("neural_renderer", cuda=cuda, projection=nr_projection, vertices_to_faces=nr_vertices_to_faces,
                lighting=nr_lighting, look_at=nr_look_at, look=nr_look, perspective=nr_perspective)
        proj = _module("libyana.camutils.project", batch_proj2d=batch_proj2d)
        camutils = _module("libyana.camutils", project=proj)
        tex = _module("libyana.renderutils.textutils", batch_vertex_textures=batch_vertex_textures)
        cat = _module("libyana.renderutils.catmesh", batch_cat_meshes=batch_cat_meshes)
        renderutils = _module("libyana.renderutils", textutils=tex, catmesh=cat)
        _module("libyana", camutils=camutils, renderutils=renderutils)
        kl = _module("kornia.losses", SSIM=object)
        kt = _module("kornia.geometry.transform", ScalePyramid=lambda: None)
        kg = _module("kornia.geometry", transform=kt)
        _module("kornia", losses=kl, geometry=kg)
        torch.cuda.FloatTensor = torch.FloatTensor  # rasterize.py:58-85, renderer.py:55-62
        torch.cuda.IntTensor = torch.IntTensor
        torch.Tensor.cuda = lambda self, *a, **kw: self  # warpbranch.py:28-47, imgflowarp.py:80-85
        _INSTALLED = True
    from meshreg.datasets import queries
    from meshreg.models import warpbranch
    from meshreg.neurender import rasterize, renderer
    from meshreg.optim import lossutils, pyramidloss
    from meshreg.warping import imgflowarp, opticalflow

    return types.SimpleNamespace(rasterize=rasterize, renderer=renderer, opticalflow=opticalflow,
                                 imgflowarp=imgflowarp, warpbranch=warpbranch, pyramidloss=pyramidloss,
                                 lossutils=lossutils, queries=queries)
