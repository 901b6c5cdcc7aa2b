"""CPU oracle for the warp half of the hot path (numpy, vectorised).

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of bench.py.  The shipped package never imports it.

PINNED: unlike the render half, the functions restated here exist in /root/reference
(meshreg/warping/imgflowarp.py, meshreg/optim/lossutils.py, pyramidloss.py:56-62) and
import on CPU; tests/golden/warp_*.npz were generated from the real reference by
tests/golden/make_golden_warp.py and this file is checked against them in
tests/test_oracle_warp.py.  The only third-party arithmetic is torch's
``grid_sample`` (zeros padding, align_corners=False), restated from the published
formula ``ix = ((x + 1) * W - 1) / 2`` (bilinear: 4 taps, weights from the opposite
corner; nearest: round-half-even).
"""
import numpy as np

F32 = np.float32


def get_spatial_meshgrid(shape, scale=False):
    """imgflowarp.get_spatial_meshgrid (imgflowarp.py:8-28) for x of shape [B,C,H,W]."""
    B, _, H, W = shape
    xx = np.broadcast_to(np.arange(W, dtype=F32)[None, :], (H, W))
    yy = np.broadcast_to(np.arange(H, dtype=F32)[:, None], (H, W))
    grid = np.stack([xx, yy], 0)[None].repeat(B, 0).astype(F32)
    if scale:
        grid[:, 0] = grid[:, 0] / F32(W)
        grid[:, 1] = grid[:, 1] / F32(H)
    return grid


def _unnormalize(coord, size):
    # align_corners=False: ((coord + 1) * size - 1) / 2, fp32 op by op
    return ((coord + F32(1)) * F32(size) - F32(1)) / F32(2)


def _gather(x, iy, ix):
    """x[B,C,H,W] sampled at integer (iy, ix)[B,H,W] with zeros padding -> [B,C,H,W], inb[B,H,W]."""
    B, C, H, W = x.shape
    inb = (ix >= 0) & (ix < W) & (iy >= 0) & (iy < H)
    ixc = np.clip(ix, 0, W - 1).astype(np.int64)
    iyc = np.clip(iy, 0, H - 1).astype(np.int64)
    b = np.arange(B)[:, None, None]
    vals = x[b, :, iyc, ixc]  # [B,H,W,C]
    vals = np.moveaxis(vals, -1, 1)
    return vals * inb[:, None].astype(F32), inb


def grid_sample(x, vgrid, mode="bilinear", with_grad=False):
    """torch.nn.functional.grid_sample(x, vgrid[B,H,W,2], mode, 'zeros', align_corners=False).

    with_grad=True additionally returns (d out / d ix, d out / d iy) per channel, in
    UNNORMALISED pixel units (bilinear only)."""
    x = np.ascontiguousarray(x, F32)
    B, C, H, W = x.shape
    ix = _unnormalize(vgrid[..., 0].astype(F32), W)
    iy = _unnormalize(vgrid[..., 1].astype(F32), H)
    if mode == "nearest":
        ixn = np.rint(ix)
        iyn = np.rint(iy)
        ok = np.isfinite(ixn) & np.isfinite(iyn)
        ixn = np.where(ok, ixn, -10).astype(np.int64)
        iyn = np.where(ok, iyn, -10).astype(np.int64)
        out, _ = _gather(x, iyn, ixn)
        return out.astype(F32)
    ix_nw = np.floor(ix)
    iy_nw = np.floor(iy)
    ix_ne, iy_ne = ix_nw + 1, iy_nw
    ix_sw, iy_sw = ix_nw, iy_nw + 1
    ix_se, iy_se = ix_nw + 1, iy_nw + 1
    nw = (ix_se - ix) * (iy_se - iy)
    ne = (ix - ix_sw) * (iy_sw - iy)
    sw = (ix_ne - ix) * (iy - iy_ne)
    se = (ix - ix_nw) * (iy - iy_nw)

    def toint(a):
        a = np.where(np.isfinite(a), a, -10)
        return np.clip(a, -10, 1 << 30).astype(np.int64)

    v_nw, _ = _gather(x, toint(iy_nw), toint(ix_nw))
    v_ne, _ = _gather(x, toint(iy_ne), toint(ix_ne))
    v_sw, _ = _gather(x, toint(iy_sw), toint(ix_sw))
    v_se, _ = _gather(x, toint(iy_se), toint(ix_se))
    out = v_nw * nw[:, None] + v_ne * ne[:, None] + v_sw * sw[:, None] + v_se * se[:, None]
    out = out.astype(F32)
    if not with_grad:
        return out
    gix = (-v_nw * (iy_se - iy)[:, None] + v_ne * (iy_sw - iy)[:, None]
           - v_sw * (iy - iy_ne)[:, None] + v_se * (iy - iy_nw)[:, None])
    giy = (-v_nw * (ix_se - ix)[:, None] - v_ne * (ix - ix_sw)[:, None]
           + v_sw * (ix_ne - ix)[:, None] + v_se * (ix - ix_nw)[:, None])
    return out, gix.astype(F32), giy.astype(F32)


def _vgrid(flow):
    """imgflowarp.py:41-48: grid + flow, normalised by (W-1), (H-1)."""
    B, _, H, W = flow.shape
    grid = get_spatial_meshgrid((B, 2, H, W))
    v = grid + flow.astype(F32)
    vx = F32(2.0) * v[:, 0] / F32(max(W - 1, 1)) - F32(1.0)
    vy = F32(2.0) * v[:, 1] / F32(max(H - 1, 1)) - F32(1.0)
    return np.stack([vx, vy], -1).astype(F32)


def warp(x, flow, thresh=0.99999, mode="bilinear"):
    """imgflowarp.warp (imgflowarp.py:31-55) -> (output * mask, mask)."""
    x = np.ascontiguousarray(x, F32)
    vgrid = _vgrid(flow)
    output = grid_sample(x, vgrid, mode)
    mask = grid_sample(np.ones_like(x), vgrid, mode)
    mask = np.where(mask < F32(thresh), F32(0), mask)
    mask = np.where(mask > 0, F32(1), mask).astype(F32)
    return (output * mask).astype(F32), mask


def warp_backward(x, flow, grad_out, thresh=0.99999):
    """Adjoint of warp (bilinear) w.r.t. flow -> [B,2,H,W] (the mask carries no gradient,
    SURVEY Q7) and w.r.t. x -> [B,C,H,W]."""
    x = np.ascontiguousarray(x, F32)
    B, C, H, W = x.shape
    vgrid = _vgrid(flow)
    _, gix, giy = grid_sample(x, vgrid, "bilinear", with_grad=True)
    _, mask = warp(x, flow, thresh)
    g = grad_out.astype(F32) * mask
    # chain: ix = ((vx+1)*W-1)/2, vx = 2 (x+u)/(W-1) - 1  =>  d ix / d u = W / (W-1)
    sx = F32(W) / F32(2) * (F32(2.0) / F32(max(W - 1, 1)))
    sy = F32(H) / F32(2) * (F32(2.0) / F32(max(H - 1, 1)))
    gflow = np.stack([(g * gix).sum(1) * sx, (g * giy).sum(1) * sy], 1).astype(F32)
    # grad wrt x: scatter of the four taps
    ix = _unnormalize(vgrid[..., 0], W)
    iy = _unnormalize(vgrid[..., 1], H)
    x0 = np.floor(ix)
    y0 = np.floor(iy)
    gx = np.zeros((B, C, H, W), np.float64)
    taps = [(0, 0, (x0 + 1 - ix) * (y0 + 1 - iy)), (1, 0, (ix - x0) * (y0 + 1 - iy)),
            (0, 1, (x0 + 1 - ix) * (iy - y0)), (1, 1, (ix - x0) * (iy - y0))]
    bidx = np.arange(B)[:, None, None].repeat(H, 1).repeat(W, 2)
    for dx, dy, w in taps:
        xi = (x0 + dx)
        yi = (y0 + dy)
        ok = np.isfinite(xi) & np.isfinite(yi) & (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
        xi_ = np.where(ok, xi, 0).astype(np.int64)
        yi_ = np.where(ok, yi, 0).astype(np.int64)
        for c in range(C):
            np.add.at(gx[:, c], (bidx[ok], yi_[ok], xi_[ok]), (g[:, c] * w)[ok])
    return gflow, gx.astype(F32)


def batch_masked_mean_loss(dists, mask):
    """lossutils.batch_masked_mean_loss (lossutils.py:1-8)."""
    mask = mask.astype(F32)
    axes = tuple(range(1, dists.ndim))
    batch_sum = (mask * dists).sum(axis=axes, dtype=np.float64)
    valid = mask.sum(axis=axes, dtype=np.float64)
    valid[valid == 0] = 1
    return (batch_sum / valid).astype(F32)


def pair_consist(recons_flow, image_ref, image, jitter_mask_ref, jitter_mask, use_backward=False,
                 thresh=0.99999):
    """imgflowarp.pair_consist (imgflowarp.py:58-115) with PyramidCriterion('l1'), level_nb=1
    (pyramidloss.py:56-62).  recons_flow = [flow12, flow21], each [B,H,W,2]."""
    f0 = recons_flow[0].transpose(0, 3, 1, 2)
    f1 = recons_flow[1].transpose(0, 3, 1, 2)
    warp1, warp_mask1 = warp(image_ref, f1, thresh)
    warpjitter1, _ = warp(jitter_mask_ref, f0, thresh)
    warp2, warp_mask2 = warp(image, f0, thresh)
    warpjitter2, _ = warp(jitter_mask, f1, thresh)
    warp_mask1 = warp_mask1 * (warpjitter2 == 1).astype(F32)
    warp_mask2 = warp_mask2 * (warpjitter1 == 1).astype(F32)
    flow_mask1 = ~(recons_flow[1] == 0)
    valid_mask1 = (warp_mask1[:, 0] != 0) & flow_mask1[..., 0] & (jitter_mask[:, 0] == 1)
    flow_mask2 = ~(recons_flow[0] == 0)
    valid_mask2 = (warp_mask2[:, 0] != 0) & flow_mask2[..., 0] & (jitter_mask_ref[:, 0] == 1)
    diff_fwd = np.abs(warp1 - image.astype(F32))
    diff_bwd = np.abs(warp2 - image_ref.astype(F32))
    m1 = np.repeat(valid_mask1[:, None], 3, 1)
    m2 = np.repeat(valid_mask2[:, None], 3, 1)
    losses_fwd = batch_masked_mean_loss(diff_fwd, m1)
    losses_bwd = batch_masked_mean_loss(diff_bwd, m2)
    warp_loss = losses_bwd + losses_fwd if use_backward else losses_fwd
    masks = [
        {"warp_mask": warp_mask1, "full_mask": valid_mask1, "flow_mask": flow_mask1},
        {"warp_mask": warp_mask2, "full_mask": valid_mask2, "flow_mask": flow_mask2},
    ]
    return warp_loss.astype(F32), masks, [warp1, warp2], [diff_fwd, diff_bwd], (losses_fwd, losses_bwd)


def pair_consist_grad(recons_flow, image_ref, image, jitter_mask_ref, jitter_mask, grad_loss,
                      use_backward=False, thresh=0.99999):
    """d (sum_b grad_loss[b] * warp_loss[b]) / d recons_flow -> [g_flow12, g_flow21], [B,H,W,2]."""
    _, masks, warps, _, _ = pair_consist(recons_flow, image_ref, image, jitter_mask_ref, jitter_mask,
                                         use_backward, thresh)
    f0 = recons_flow[0].transpose(0, 3, 1, 2)
    f1 = recons_flow[1].transpose(0, 3, 1, 2)
    B = image.shape[0]
    out = []
    for flow, src, tgt, warped, m in (
        (f1, image_ref, image, warps[0], masks[0]["full_mask"]),
        (f0, image, image_ref, warps[1], masks[1]["full_mask"]),
    ):
        cnt = np.maximum(3.0 * m.reshape(B, -1).sum(1), 1.0)
        cnt[m.reshape(B, -1).sum(1) == 0] = 1
        g = np.sign(warped - tgt.astype(F32)) * m[:, None].astype(F32)
        g = g * (grad_loss.astype(F32) / cnt.astype(F32))[:, None, None, None]
        gflow, _ = warp_backward(src, flow, g.astype(F32), thresh)
        out.append(gflow.transpose(0, 2, 3, 1).astype(F32))
    g21, g12 = out
    if not use_backward:
        g12 = np.zeros_like(g12)
    return [g12, g21]


def occlusion_mask_from_warped_grid(grid, warped_grid, distance_thresh=0.03):
    """imgflowarp.occlusion_mask_from_warped_grid (imgflowarp.py:149-172); [B,H,W,4] inputs."""
    mask = grid[..., 2] * warped_grid[..., 2]
    d = ((warped_grid - grid) * mask[..., None])[..., :2].astype(F32)
    displ = np.sqrt((d * d).sum(-1, dtype=F32))
    motion = (displ < F32(distance_thresh)).astype(F32)
    return (mask * motion).astype(F32)


def get_occlusion_mask(mask_flow1, mask_flow2, flow12, flow21, thresh=0.99999):
    """imgflowarp.get_occlusion_mask (imgflowarp.py:118-146).  mask_flow*[B,1,H,W],
    flow*[B,3,H,W] -> occl_mask1, occl_mask2 [B,H,W]."""
    shp = mask_flow1.shape
    grid1 = np.concatenate([get_spatial_meshgrid(shp, True), mask_flow1, mask_flow1], 1).astype(F32)
    grid2 = np.concatenate([get_spatial_meshgrid(shp, True), mask_flow2, mask_flow2], 1).astype(F32)
    warp_grid12, _ = warp(grid1, flow21[:, :2], thresh, "nearest")
    warp_grid21, _ = warp(grid2, flow12[:, :2], thresh, "nearest")
    warp_grid12 = warp_grid12 * mask_flow2[:, :1]
    warp_grid21 = warp_grid21 * mask_flow1[:, :1]
    warp_grid212, _ = warp(warp_grid12, flow12[:, :2], thresh, "nearest")
    warp_grid121, _ = warp(warp_grid21, flow21[:, :2], thresh, "nearest")
    warp_grid212 = warp_grid212 * mask_flow1[:, :1]
    warp_grid121 = warp_grid121 * mask_flow2[:, :1]
    occl1 = occlusion_mask_from_warped_grid(grid1.transpose(0, 2, 3, 1), warp_grid212.transpose(0, 2, 3, 1))
    occl2 = occlusion_mask_from_warped_grid(grid2.transpose(0, 2, 3, 1), warp_grid121.transpose(0, 2, 3, 1))
    return occl1, occl2


def get_opticalflow(raster, verts_cam, faces_idx, camintrs, renderer_kw, orig_img_size=None,
                    mask_occlusions=True, ignore_face_idxs=None, return_renders=False):
    """opticalflow.get_opticalflow (opticalflow.py:51-156) on top of the render oracle.
    `raster` is the oracle.raster_ref module; renderer_kw are the Renderer settings of
    warpreg.py:40-51 (image_size, R, t, dist_coeffs, orig_size, near, far, eps...)."""
    loc1 = raster.batch_proj2d(verts_cam[0], camintrs[0])
    loc2 = raster.batch_proj2d(verts_cam[1], camintrs[1])
    outs = []
    for (v, K, displ) in ((verts_cam[0], camintrs[0], loc2 - loc1), (verts_cam[1], camintrs[1], loc1 - loc2)):
        sample_flows = np.concatenate([displ, np.ones_like(displ[:, :, :1])], -1)
        tex = raster.batch_vertex_textures(faces_idx, sample_flows)
        ro = raster.render(v, faces_idx, tex, K, **renderer_kw)
        mask = (ro["alpha"][:, None] > F32(0.99999)).astype(F32)
        if ignore_face_idxs is not None:
            fim = ro["face_index_map"]
            ign = np.abs(fim[..., None] - np.asarray(ignore_face_idxs, np.int32)).min(-1) != 0
            ign = ign[:, ::-1]
            mask = mask * ign.astype(F32)[:, None]
        outs.append((ro, mask, ro["rgb"] * mask))
    (ro1, mask_flow1, pred12), (ro2, mask_flow2, pred21) = outs
    if mask_occlusions:
        mask_flow2 = ro2["alpha"][:, None].astype(F32)  # raw alpha
        occl1, occl2 = get_occlusion_mask(mask_flow1, mask_flow2, pred12, pred21)
        mask_flow1 = mask_flow1 * occl1[:, None]
        mask_flow2 = mask_flow2 * occl2[:, None]
        pred12 = pred12 * mask_flow1
        pred21 = pred21 * mask_flow2
    pred12 = pred12.transpose(0, 2, 3, 1)[..., :2]
    pred21 = pred21.transpose(0, 2, 3, 1)[..., :2]
    if orig_img_size is not None:
        pred12 = pred12[:, : orig_img_size[1], : orig_img_size[0]]
        pred21 = pred21[:, : orig_img_size[1], : orig_img_size[0]]
    flows = [np.ascontiguousarray(pred12, F32), np.ascontiguousarray(pred21, F32)]
    if return_renders:
        return flows, [ro1, ro2]
    return flows
