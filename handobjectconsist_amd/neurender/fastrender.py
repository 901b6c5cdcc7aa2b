"""Lit RGBA render of a posed mesh -- counterpart of ``render`` in meshreg/neurender/fastrender.py:14-59
(SURVEY 8f "f3"; the reference uses it for visualisation).  The composite-figure helpers of that file
(``comp_render`` and friends: matplotlib colour maps, rotated views, figure layout) are out of scope.
"""
import torch

from handobjectconsist_amd.neurender import renderer
from handobjectconsist_amd.utils import textutils


def render(verts, faces, input_res, camintrs=None, colors=None, fill_back=True, near=0.05, far=2, crop_to_img=True,
           bg_color=None):
    """verts [B,V,3] (camera frame), faces [B,F,3], input_res (W, H), camintrs [B,3,3], colors [B,V,>=3]
    -> [B,H,W,4] RGBA: vertex colours under the renderer's directional light (ambient 0.8), alpha = coverage,
    optionally composited over `bg_color`.  Rendered on a square raster of max(input_res) and cropped to the
    top-left H x W like the reference."""
    side = max(input_res)
    dev = verts.device
    neurenderer = renderer.Renderer(
        image_size=side, orig_size=side, K=camintrs, R=torch.eye(3, device=dev).unsqueeze(0),
        t=torch.zeros(1, 3, device=dev), anti_aliasing=False, fill_back=fill_back, near=near, far=far, no_light=False,
        light_intensity_ambient=0.8)
    if colors is None:
        colors = torch.ones_like(verts)
    out = neurenderer(verts, faces, textutils.batch_vertex_textures(faces, colors[:, :, :3]))
    rgb, alpha = out["rgb"], out["alpha"]
    if crop_to_img:
        width, height = input_res[0], input_res[1]
        rgb, alpha = rgb[:, :, :height, :width], alpha[:, :height, :width]
    if bg_color is not None:
        # (fastrender.py:57 multiplies [B,3,H,W] by [B,H,W]: that broadcasts for B = 1 only -- B = 2 raises, B = 3 would
        # take the alpha of SAMPLE c for channel c.  Per-sample alpha here: a conscious fix, equal to the reference
        # wherever it runs; tests/test_gpu_chain.py::test_fastrender_render_against_the_reference)
        a = alpha.unsqueeze(1)
        rgb = rgb * a + bg_color * (1 - a) * torch.ones_like(rgb)
    return torch.cat([rgb, alpha.unsqueeze(1)], 1).permute(0, 2, 3, 1)
