"""``Renderer`` -- drop-in for meshreg/neurender/renderer.py (reference renderer.py:12-295).

Same constructor arguments, same ``forward`` / ``render`` / ``render_rgb`` /
``render_silhouettes`` / ``render_depth`` / ``project`` signatures and the same quirks
(``rasterizer_eps = 1e-3`` is only passed by ``render`` / ``render_rgb``; silhouettes and
depth use the module default 1e-4 -- SURVEY Q1).  The camera / fill-back / gather steps
are small differentiable PyTorch programs (nr_ops.py); rasterisation goes to the HIP
kernels through rasterize.py.
"""
from __future__ import division

import math

import numpy
import torch
import torch.nn as nn

from handobjectconsist_amd.neurender import nr_ops as nr
from handobjectconsist_amd.neurender import rasterize


def _device():
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


class Renderer(nn.Module):
    def __init__(
        self,
        image_size=256,
        anti_aliasing=True,
        background_color=[0, 0, 0],
        fill_back=True,
        camera_mode="projection",
        K=None,
        R=None,
        t=None,
        dist_coeffs=None,
        orig_size=1024,
        perspective=True,
        viewing_angle=30,
        camera_direction=[0, 0, 1],
        near=0.1,
        far=100,
        light_intensity_ambient=0.5,
        light_intensity_directional=0.5,
        light_color_ambient=[1, 1, 1],
        light_color_directional=[1, 1, 1],
        light_direction=[0, 1, 0],
        no_light=False,
    ):
        """
        Wrapper on top of the rasteriser (reference renderer.py:13-83)
        """
        super(Renderer, self).__init__()
        # rendering
        self.image_size = image_size
        self.anti_aliasing = anti_aliasing
        self.background_color = background_color
        self.fill_back = fill_back
        self.no_light = no_light

        # camera
        self.camera_mode = camera_mode
        if self.camera_mode == "projection":
            self.K = K
            self.R = R
            self.t = t
            dev = _device()
            if isinstance(self.K, numpy.ndarray):
                self.K = torch.tensor(self.K, dtype=torch.float32, device=dev)
            if isinstance(self.R, numpy.ndarray):
                self.R = torch.tensor(self.R, dtype=torch.float32, device=dev)
            if isinstance(self.t, numpy.ndarray):
                self.t = torch.tensor(self.t, dtype=torch.float32, device=dev)
            self.dist_coeffs = dist_coeffs
            if dist_coeffs is None:
                self.dist_coeffs = torch.tensor([[0.0, 0.0, 0.0, 0.0, 0.0]], dtype=torch.float32, device=dev)
            self.orig_size = orig_size
        elif self.camera_mode in ["look", "look_at"]:
            self.perspective = perspective
            self.viewing_angle = viewing_angle
            self.eye = [0, 0, -(1.0 / math.tan(math.radians(self.viewing_angle)) + 1)]
            self.camera_direction = [0, 0, 1]
        else:
            raise ValueError("Camera mode has to be one of projection, look or look_at")

        self.near = near
        self.far = far

        # light
        self.light_intensity_ambient = light_intensity_ambient
        self.light_intensity_directional = light_intensity_directional
        self.light_color_ambient = light_color_ambient
        self.light_color_directional = light_color_directional
        self.light_direction = light_direction

        # rasterization
        self.rasterizer_eps = 1e-3

    def forward(
        self,
        vertices,
        faces,
        textures=None,
        mode=None,
        K=None,
        R=None,
        t=None,
        dist_coeffs=None,
        orig_size=None,
        detach_renders=False,
    ):
        """
        Implementation of forward rendering method (reference renderer.py:85-114)
        """
        if mode is None:
            return self.render(
                vertices, faces, textures, K, R, t, dist_coeffs, orig_size, detach_renders=detach_renders
            )
        elif mode == "rgb":
            return self.render_rgb(vertices, faces, textures, K, R, t, dist_coeffs, orig_size)
        elif mode == "silhouettes":
            return self.render_silhouettes(vertices, faces, K, R, t, dist_coeffs, orig_size)
        elif mode == "depth":
            return self.render_depth(vertices, faces, K, R, t, dist_coeffs, orig_size)
        else:
            raise ValueError("mode should be one of None, 'silhouettes' or 'depth'")

    # -- helpers ---------------------------------------------------------------------------
    @staticmethod
    def _fill_back_faces(faces):
        # renderer.py:251: cat(faces, faces[:, :, ::-1])
        return torch.cat((faces, faces.flip(-1)), dim=1).detach()

    @staticmethod
    def _fill_back_textures(textures):
        # renderer.py:252
        return torch.cat((textures, textures.permute((0, 1, 4, 3, 2, 5))), dim=1)

    def _light(self, vertices, faces, textures):
        faces_lighting = nr.vertices_to_faces(vertices, faces)
        return nr.lighting(
            faces_lighting,
            textures,
            self.light_intensity_ambient,
            self.light_intensity_directional,
            self.light_color_ambient,
            self.light_color_directional,
            self.light_direction,
        )

    # -- modes -----------------------------------------------------------------------------
    def render_silhouettes(self, vertices, faces, K=None, R=None, t=None, dist_coeffs=None, orig_size=None):
        if self.fill_back:
            faces = self._fill_back_faces(faces)
        vertices = self.project(vertices, K=K, R=R, t=t, dist_coeffs=dist_coeffs, orig_size=orig_size)
        faces = nr.vertices_to_faces(vertices, faces)
        images = rasterize.rasterize_silhouettes(faces, self.image_size, self.anti_aliasing)
        return images

    def render_depth(self, vertices, faces, K=None, R=None, t=None, dist_coeffs=None, orig_size=None):
        if self.fill_back:
            faces = self._fill_back_faces(faces)
        vertices = self.project(vertices, K=K, R=R, t=t, dist_coeffs=dist_coeffs, orig_size=orig_size)
        faces = nr.vertices_to_faces(vertices, faces)
        images = rasterize.rasterize_depth(faces, self.image_size, self.anti_aliasing)
        return images

    def project(self, vertices, K=None, R=None, t=None, dist_coeffs=None, orig_size=None):
        # viewpoint transformation (reference renderer.py:164-188)
        if self.camera_mode == "look_at":
            vertices = nr.look_at(vertices, self.eye)
            if self.perspective:
                vertices = nr.perspective(vertices, angle=self.viewing_angle)
        elif self.camera_mode == "look":
            vertices = nr.look(vertices, self.eye, self.camera_direction)
            if self.perspective:
                vertices = nr.perspective(vertices, angle=self.viewing_angle)
        elif self.camera_mode == "projection":
            if K is None:
                K = self.K
            if R is None:
                R = self.R
            if t is None:
                t = self.t
            if dist_coeffs is None:
                dist_coeffs = self.dist_coeffs
            if orig_size is None:
                orig_size = self.orig_size
            dev = vertices.device
            vertices = nr.projection(vertices, K.to(dev), R.to(dev), t.to(dev), dist_coeffs.to(dev), orig_size)
        return vertices

    def render_rgb(self, vertices, faces, textures, K=None, R=None, t=None, dist_coeffs=None, orig_size=None):
        if self.fill_back:
            faces = self._fill_back_faces(faces)
            textures = self._fill_back_textures(textures)
        if not self.no_light:
            textures = self._light(vertices, faces, textures)
        vertices = self.project(vertices, K=K, R=R, t=t, dist_coeffs=dist_coeffs, orig_size=orig_size)
        faces = nr.vertices_to_faces(vertices, faces)
        images = rasterize.rasterize(
            faces,
            textures,
            self.image_size,
            self.anti_aliasing,
            self.near,
            self.far,
            self.rasterizer_eps,
            self.background_color,
        )
        return images

    def render_vertex_colors(self, vertices, faces, vertex_colors, K=None, R=None, t=None, dist_coeffs=None,
                             orig_size=None):
        """``render(vertices, faces, batch_vertex_textures(faces, vertex_colors), ...,
        detach_renders=True)`` through the fused vertex-colour kernels: same dict, same values,
        gradient w.r.t. ``vertex_colors`` only.  Used by opticalflow.get_opticalflow (the training
        path renders exactly this: opticalflow.py:101-108 with detach_renders=True).  Not available
        with lighting (``no_light=False``)."""
        if not self.no_light:
            raise ValueError("render_vertex_colors requires no_light=True")
        v = self.project(vertices, K=K, R=R, t=t, dist_coeffs=dist_coeffs, orig_size=orig_size).detach()
        return rasterize.rasterize_vertex_colors(
            v, faces, vertex_colors, self.fill_back, self.image_size, self.anti_aliasing, self.near, self.far,
            self.rasterizer_eps, self.background_color)

    def render(
        self,
        vertices,
        faces,
        textures,
        K=None,
        R=None,
        t=None,
        dist_coeffs=None,
        orig_size=None,
        detach_renders=False,
    ):
        """rgb + alpha + depth + index/weight maps as a dict (reference renderer.py:237-295)."""
        if self.fill_back:
            faces = self._fill_back_faces(faces)
            textures = self._fill_back_textures(textures)
        if not self.no_light:
            textures = self._light(vertices, faces, textures)
        vertices = self.project(vertices, K=K, R=R, t=t, dist_coeffs=dist_coeffs, orig_size=orig_size)
        faces = nr.vertices_to_faces(vertices, faces)
        if detach_renders:
            faces = faces.detach()
        out = rasterize.rasterize_rgbad(
            faces,
            textures,
            self.image_size,
            self.anti_aliasing,
            self.near,
            self.far,
            self.rasterizer_eps,
            self.background_color,
        )
        return out
