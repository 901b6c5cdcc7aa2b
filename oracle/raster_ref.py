"""CPU oracle for the render half of the hot path (numpy + the C restatement).

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of bench.py.  The shipped package never imports it.

PARITY UNPINNED for everything in this file: the arithmetic restated here lives in
third-party packages absent from /root/reference and from this image
(``neural_renderer`` un-pinned, ``libyana@v0.2.0``; /root/reference/environment.yml:35-36).
What IS pinned by the reference are the call-site contracts, cited per function.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

DEFAULT_NEAR = 0.1
DEFAULT_FAR = 100
DEFAULT_EPS = 1e-4


def build(force=False):
    """Compile oracle/raster_oracle.c -> oracle/liboracle_raster.so (gcc, see Makefile)."""
    so = os.path.join(_HERE, "liboracle_raster.so")
    src = os.path.join(_HERE, "raster_oracle.c")
    if force or not os.path.exists(so) or (
        os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.oracle_max_threads.restype = ctypes.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---------------------------------------------------------------------------------------
# The five native entry points, with the reference's buffer conventions
# ---------------------------------------------------------------------------------------


def forward_face_index_map(faces, image_size, near, far, return_depth=True, num_threads=1):
    """Kernels A+B with the pre-fills of rasterize.py:60-85 / :201."""
    faces = _f32(faces)
    B, F = faces.shape[:2]
    is_ = int(image_size)
    fim = np.full((B, is_, is_), -1, np.int32)
    wmap = np.zeros((B, is_, is_, 3), np.float32)
    dmap = np.full((B, is_, is_), far, np.float32)
    finv_map = np.zeros((B, is_, is_, 3, 3) if return_depth else (1,), np.float32)
    faces_inv = np.zeros_like(faces)
    lib().oracle_forward_face_index_map(
        _p(faces), _p(fim), _p(wmap), _p(dmap), _p(finv_map), _p(faces_inv),
        ctypes.c_int(B), ctypes.c_int(F), ctypes.c_int(is_), ctypes.c_float(near),
        ctypes.c_float(far), ctypes.c_int(1), ctypes.c_int(1), ctypes.c_int(int(return_depth)),
        ctypes.c_int(num_threads),
    )
    return fim, wmap, dmap, finv_map, faces_inv


def forward_texture_sampling(faces, textures, fim, wmap, dmap, eps):
    """Kernel C with the pre-fills of rasterize.py:64-71."""
    faces, textures = _f32(faces), _f32(textures)
    B, F = faces.shape[:2]
    is_ = fim.shape[1]
    ts = textures.shape[2]
    rgb = np.zeros((B, is_, is_, 3), np.float32)
    sidx = np.zeros((B, is_, is_, 8), np.int32)
    swgt = np.zeros((B, is_, is_, 8), np.float32)
    lib().oracle_forward_texture_sampling(
        _p(faces), _p(textures), _p(fim), _p(wmap), _p(dmap), _p(rgb), _p(sidx), _p(swgt),
        ctypes.c_int(B), ctypes.c_int(F), ctypes.c_int(is_), ctypes.c_int(ts), ctypes.c_float(eps),
    )
    return rgb, sidx, swgt


def backward_pixel_map(faces, fim, rgb, alpha, grad_rgb, grad_alpha, eps, return_rgb, return_alpha,
                       num_threads=1):
    faces = _f32(faces)
    B, F = faces.shape[:2]
    is_ = fim.shape[1]
    grad_faces = np.zeros_like(faces)
    dummy = np.zeros((1,), np.float32)
    lib().oracle_backward_pixel_map(
        _p(faces), _p(fim), _p(_f32(rgb) if return_rgb else dummy),
        _p(_f32(alpha) if return_alpha else dummy),
        _p(_f32(grad_rgb) if return_rgb else dummy),
        _p(_f32(grad_alpha) if return_alpha else dummy), _p(grad_faces),
        ctypes.c_int(B), ctypes.c_int(F), ctypes.c_int(is_), ctypes.c_float(eps),
        ctypes.c_int(int(return_rgb)), ctypes.c_int(int(return_alpha)), ctypes.c_int(num_threads),
    )
    return grad_faces


def backward_textures(fim, swgt, sidx, grad_rgb, num_faces, texture_size):
    B, is_ = fim.shape[:2]
    ts = texture_size
    grad_textures = np.zeros((B, num_faces, ts, ts, ts, 3), np.float32)
    lib().oracle_backward_textures(
        _p(fim), _p(_f32(swgt)), _p(np.ascontiguousarray(sidx, np.int32)), _p(_f32(grad_rgb)),
        _p(grad_textures), ctypes.c_int(B), ctypes.c_int(num_faces), ctypes.c_int(is_),
        ctypes.c_int(ts),
    )
    return grad_textures


def backward_depth_map(faces, dmap, fim, finv_map, wmap, grad_depth, grad_faces):
    faces = _f32(faces)
    B, F = faces.shape[:2]
    is_ = fim.shape[1]
    grad_faces = _f32(grad_faces).copy()
    lib().oracle_backward_depth_map(
        _p(faces), _p(_f32(dmap)), _p(fim), _p(_f32(finv_map)), _p(_f32(wmap)),
        _p(_f32(grad_depth)), _p(grad_faces), ctypes.c_int(B), ctypes.c_int(F), ctypes.c_int(is_),
    )
    return grad_faces


# ---------------------------------------------------------------------------------------
# RasterizeFunction / rasterize_rgbad (rasterize.py:16-448), numpy
# ---------------------------------------------------------------------------------------


def rasterize_forward(faces, textures, image_size, near, far, eps, background_color,
                      return_rgb=True, return_alpha=True, return_depth=True, num_threads=1):
    """RasterizeFunction.forward (rasterize.py:23-125).  Returns the raw (RASTER
    orientation, NHWC) maps plus everything backward() needs."""
    faces = _f32(faces)
    fim, wmap, dmap, finv_map, faces_inv = forward_face_index_map(
        faces, image_size, near, far, return_depth, num_threads)
    out = dict(faces=faces, face_index_map=fim, weight_map=wmap, depth_map=dmap,
               face_inv_map=finv_map, faces_inv=faces_inv, image_size=image_size, eps=eps,
               return_rgb=return_rgb, return_alpha=return_alpha, return_depth=return_depth)
    if return_rgb:
        textures = _f32(textures)
        rgb, sidx, swgt = forward_texture_sampling(faces, textures, fim, wmap, dmap, eps)
        # forward_background, rasterize.py:251-260
        bg = np.asarray(background_color, np.float32)
        mask = (fim >= 0).astype(np.float32)[..., None]
        if bg.ndim == 1:
            rgb = rgb * mask + (1 - mask) * bg[None, None, None, :]
        else:
            rgb = rgb * mask + (1 - mask) * bg[:, None, None, :]
        out.update(textures=textures, rgb_map=rgb.astype(np.float32), sampling_index_map=sidx,
                   sampling_weight_map=swgt)
    if return_alpha:
        # forward_alpha_map, rasterize.py:245-248
        out["alpha_map"] = (fim >= 0).astype(np.float32)
    return out


def rasterize_backward(saved, grad_rgb_map=None, grad_alpha_map=None, grad_depth_map=None,
                       num_threads=1):
    """RasterizeFunction.backward (rasterize.py:127-197); grads in RASTER orientation NHWC."""
    faces, fim = saved["faces"], saved["face_index_map"]
    B, F = faces.shape[:2]
    rr, ra, rd = saved["return_rgb"], saved["return_alpha"], saved["return_depth"]
    grad_faces = np.zeros_like(faces)
    grad_textures = None
    if rr and grad_rgb_map is None:
        grad_rgb_map = np.zeros_like(saved["rgb_map"])
    if ra and grad_alpha_map is None:
        grad_alpha_map = np.zeros_like(saved["alpha_map"])
    if rd and grad_depth_map is None:
        grad_depth_map = np.zeros_like(saved["depth_map"])
    if rr or ra:
        grad_faces = backward_pixel_map(
            faces, fim, saved.get("rgb_map"), saved.get("alpha_map"), grad_rgb_map, grad_alpha_map,
            saved["eps"], rr, ra, num_threads)
    if rr:
        ts = saved["textures"].shape[2]
        grad_textures = backward_textures(
            fim, saved["sampling_weight_map"], saved["sampling_index_map"], grad_rgb_map, F, ts)
    if rd:
        grad_faces = backward_depth_map(
            faces, saved["depth_map"], fim, saved["face_inv_map"], saved["weight_map"],
            grad_depth_map, grad_faces)
    return grad_faces, grad_textures


def _avg_pool2(x):
    """F.avg_pool2d(kernel_size=2) over the last two dims."""
    a = x[..., 0::2, 0::2]
    b = x[..., 0::2, 1::2]
    c = x[..., 1::2, 0::2]
    d = x[..., 1::2, 1::2]
    return ((a + b + c + d) * np.float32(0.25)).astype(np.float32)


def rasterize_rgbad(faces, textures=None, image_size=256, anti_aliasing=True, near=DEFAULT_NEAR,
                    far=DEFAULT_FAR, eps=DEFAULT_EPS, background_color=(0, 0, 0), return_rgb=True,
                    return_alpha=True, return_depth=True, num_threads=1, keep_saved=False):
    """rasterize_rgbad (rasterize.py:362-448): x2 raster if AA, NCHW + vertical flip of
    rgb/alpha/depth, 2x2 average pool if AA; index/weight/inverse maps stay un-flipped."""
    is_ = image_size * 2 if anti_aliasing else image_size
    saved = rasterize_forward(faces, textures, is_, near, far, eps,
                              background_color if background_color is not None else (0, 0, 0),
                              return_rgb, return_alpha, return_depth, num_threads)
    rgb = alpha = depth = None
    if return_rgb:
        rgb = saved["rgb_map"].transpose(0, 3, 1, 2)[:, :, ::-1, :]
    if return_alpha:
        alpha = saved["alpha_map"][:, ::-1, :]
    if return_depth:
        depth = saved["depth_map"][:, ::-1, :]
    if anti_aliasing:
        rgb = _avg_pool2(rgb) if return_rgb else None
        alpha = _avg_pool2(alpha) if return_alpha else None
        depth = _avg_pool2(depth) if return_depth else None
    ret = {
        "rgb": np.ascontiguousarray(rgb) if return_rgb else None,
        "alpha": np.ascontiguousarray(alpha) if return_alpha else None,
        "depth": np.ascontiguousarray(depth) if return_depth else None,
        "face_inv_map": saved["face_inv_map"],
        "face_index_map": saved["face_index_map"],
        "weight_map": saved["weight_map"],
    }
    if keep_saved:
        ret["_saved"] = saved
    return ret


# ---------------------------------------------------------------------------------------
# neural_renderer python helpers used by renderer.py (third-party, restated)
# ---------------------------------------------------------------------------------------


def nr_projection(vertices, K, R, t, dist_coeffs, orig_size, eps=1e-9):
    """nr.projection as called at renderer.py:187.  fp32 throughout."""
    v = _f32(vertices)
    K, R, t, dc = _f32(K), _f32(R), _f32(t), _f32(dist_coeffs)
    v = np.matmul(v, R.transpose(0, 2, 1)) + t.reshape(-1, 1, 3)
    x, y, z = v[:, :, 0], v[:, :, 1], v[:, :, 2]
    x_ = x / (z + np.float32(eps))
    y_ = y / (z + np.float32(eps))
    k1, k2, p1, p2, k3 = (dc[:, None, i] for i in range(5))
    r = np.sqrt(x_ ** 2 + y_ ** 2)
    x__ = x_ * (1 + k1 * (r ** 2) + k2 * (r ** 4) + k3 * (r ** 6)) + 2 * p1 * x_ * y_ + p2 * (r ** 2 + 2 * x_ ** 2)
    y__ = y_ * (1 + k1 * (r ** 2) + k2 * (r ** 4) + k3 * (r ** 6)) + p1 * (r ** 2 + 2 * y_ ** 2) + 2 * p2 * x_ * y_
    vert = np.stack([x__, y__, np.ones_like(z)], axis=-1).astype(np.float32)
    vert = np.matmul(vert, K.transpose(0, 2, 1))
    u, vv = vert[:, :, 0], vert[:, :, 1]
    vv = np.float32(orig_size) - vv
    u = 2 * (u - np.float32(orig_size) / 2.0) / np.float32(orig_size)
    vv = 2 * (vv - np.float32(orig_size) / 2.0) / np.float32(orig_size)
    return np.stack([u, vv, z], axis=-1).astype(np.float32)


def nr_vertices_to_faces(vertices, faces):
    """nr.vertices_to_faces (renderer.py:282): vertices[b, faces[b, f, k]]."""
    B = vertices.shape[0]
    faces = np.asarray(faces)
    return np.stack([vertices[b][faces[b]] for b in range(B)]).astype(np.float32)


def fill_back(faces_idx, textures=None):
    """renderer.py:250-252."""
    f2 = np.concatenate([faces_idx, faces_idx[:, :, ::-1]], axis=1)
    if textures is None:
        return f2, None
    t2 = np.concatenate([textures, textures.transpose(0, 1, 4, 3, 2, 5)], axis=1)
    return f2, np.ascontiguousarray(t2, np.float32)


def nr_lighting(faces, textures, intensity_ambient=0.5, intensity_directional=0.5,
                color_ambient=(1, 1, 1), color_directional=(1, 1, 1), direction=(0, 1, 0)):
    """nr.lighting (renderer.py:257-265)."""
    faces, textures = _f32(faces), _f32(textures)
    B, F = faces.shape[:2]
    ca = np.asarray(color_ambient, np.float32)
    cd = np.asarray(color_directional, np.float32)
    d = np.asarray(direction, np.float32)
    light = np.zeros((B, F, 3), np.float32)
    if intensity_ambient != 0:
        light = light + np.float32(intensity_ambient) * ca[None, None, :]
    if intensity_directional != 0:
        v10 = faces[:, :, 0] - faces[:, :, 1]
        v12 = faces[:, :, 2] - faces[:, :, 1]
        n = np.cross(v10, v12)
        n = n / np.maximum(np.linalg.norm(n, axis=-1, keepdims=True), 1e-5).astype(np.float32)
        cos = np.maximum((n * d[None, None, :]).sum(-1), 0).astype(np.float32)
        light = light + np.float32(intensity_directional) * (cd[None, None, :] * cos[:, :, None])
    return (textures * light[:, :, None, None, None, :]).astype(np.float32)


def render(vertices, faces_idx, textures, K, R, t, dist_coeffs, orig_size, image_size,
           anti_aliasing=False, fill_back_=True, near=0.1, far=100, eps=1e-3,
           background_color=(0, 0, 0), num_threads=1, keep_saved=False):
    """Renderer.render with camera_mode='projection', no_light=True (renderer.py:237-295)."""
    tex = textures
    if fill_back_:
        faces_idx, tex = fill_back(faces_idx, textures)
    v = nr_projection(vertices, K, R, t, dist_coeffs, orig_size)
    faces = nr_vertices_to_faces(v, faces_idx)
    out = rasterize_rgbad(faces, tex, image_size, anti_aliasing, near, far, eps, background_color,
                          num_threads=num_threads, keep_saved=keep_saved)
    out["_faces"] = faces
    out["_textures"] = tex
    return out


# ---------------------------------------------------------------------------------------
# libyana helpers on the path (third-party, restated; SURVEY B.11 -- ASSUMED layouts)
# ---------------------------------------------------------------------------------------


def batch_proj2d(verts, camintr):
    """libyana.camutils.project.batch_proj2d (opticalflow.py:98-99)."""
    h = np.matmul(_f32(camintr), _f32(verts).transpose(0, 2, 1)).transpose(0, 2, 1)
    return (h[:, :, :2] / h[:, :, 2:]).astype(np.float32)


def batch_vertex_textures(faces_idx, vertex_colors):
    """libyana.renderutils.textutils.batch_vertex_textures (opticalflow.py:103,123):
    [B,F,2,2,2,3] with the three vertex colours at texels (1,0,0), (0,1,0), (0,0,1)."""
    B, F = faces_idx.shape[:2]
    tex = np.zeros((B, F, 2, 2, 2, 3), np.float32)
    vc = _f32(vertex_colors)
    for b in range(B):
        tex[b, :, 1, 0, 0] = vc[b][faces_idx[b, :, 0]]
        tex[b, :, 0, 1, 0] = vc[b][faces_idx[b, :, 1]]
        tex[b, :, 0, 0, 1] = vc[b][faces_idx[b, :, 2]]
    return tex


def batch_cat_meshes(verts_list, faces_list):
    """libyana.renderutils.catmesh.batch_cat_meshes (warpbranch.py:50)."""
    off, faces = 0, []
    for v, f in zip(verts_list, faces_list):
        faces.append(np.asarray(f) + off)
        off += v.shape[1]
    return np.concatenate(verts_list, 1), np.concatenate(faces, 1), None
