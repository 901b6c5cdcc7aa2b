"""Flow-based image warping on MI355X -- drop-in for meshreg/warping/imgflowarp.py.

Same five public functions, signatures and return values as the reference
(/root/reference/meshreg/warping/imgflowarp.py:8, :31, :58, :118, :149).  The PyTorch op
chains of the reference (meshgrid rebuild + H2D copy per call, two ``grid_sample`` per
warp, ~15 element-wise passes per pair) are replaced by the HIP kernels of csrc/warp.hip:

* ``warp``               -> mr_warp_forward / mr_warp_backward (sample + validity in one pass)
* ``get_occlusion_mask`` -> mr_occlusion_mask (4 chained nearest warps fused, no intermediates)
* ``pair_consist``       -> mr_pair_consist_forward / _backward when the criterion is the
                            reference's default ``PyramidCriterion('l1')`` with ``level_nb=1``;
                            any other criterion goes through the composed ``warp`` path with
                            the reference's exact control flow.

Semantics preserved on purpose: Q5 (flow validity looks at the x
component only), Q6 (jitter masks warped with the opposite flow, compared ``== 1``), Q7
((W-1)-normalised grid sampled with align_corners=False; the mask carries no gradient).
"""
import torch

from handobjectconsist_amd import _lib

_GRID_CACHE = {}


def get_spatial_meshgrid(x: torch.Tensor, scale=False):
    """
    Get grid which contains spatial coordinates at each pixel location

    Args:
        x: image of shape [batch_size, channels, height, width] for which
            we want to generate the spatial grid
    """
    batch_size, _, height, width = x.size()
    key = (height, width, bool(scale), str(x.device))
    base = _GRID_CACHE.get(key)
    if base is None:
        # built once per (H, W, device) instead of at every call (Q8); on the host like the
        # reference so that the scaled variant has the host's correctly rounded divisions
        xx = torch.arange(0, width).view(1, -1).repeat(height, 1)
        yy = torch.arange(0, height).view(-1, 1).repeat(1, width)
        base = torch.stack((xx, yy), 0).float()
        if scale:
            base[0] = base[0] / width
            base[1] = base[1] / height
        base = base.to(x.device)
        _GRID_CACHE[key] = base
    return base.unsqueeze(0).repeat(batch_size, 1, 1, 1)


class _WarpFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, flow, thresh, mode):
        ctx.set_materialize_grads(False)
        _lib.check_cuda(x, flow)
        x_c, flow_c = _lib.contig(x), _lib.contig(flow)
        B, C, H, W = x_c.shape
        if flow_c.shape != (B, 2, H, W):
            raise ValueError(f"flow must be [{B}, 2, {H}, {W}], got {tuple(flow_c.shape)}")
        out = torch.empty_like(x_c)
        mask = torch.empty_like(x_c)
        _lib.call("mr_warp_forward", _lib.ptr(x_c), _lib.ptr(flow_c), _lib.ptr(out), _lib.ptr(mask), B, C, H, W,
                  float(thresh), mode, _lib.stream_ptr(x_c.device))
        ctx.save_for_backward(x_c, flow_c)
        ctx.cfg = (float(thresh), mode)
        ctx.mark_non_differentiable(mask)
        return out, mask

    @staticmethod
    def backward(ctx, grad_out, _grad_mask):
        x_c, flow_c = ctx.saved_tensors
        thresh, mode = ctx.cfg
        B, C, H, W = x_c.shape
        need_x, need_flow = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if grad_out is None or not (need_x or need_flow):
            return None, None, None, None
        g = _lib.contig(grad_out)
        grad_x = torch.zeros_like(x_c) if need_x else None
        grad_flow = torch.empty_like(flow_c) if need_flow else None
        _lib.call("mr_warp_backward", _lib.ptr(x_c), _lib.ptr(flow_c), _lib.ptr(g), _lib.ptr(grad_x),
                  _lib.ptr(grad_flow), B, C, H, W, thresh, mode, _lib.stream_ptr(x_c.device))
        return grad_x, grad_flow, None, None


def warp(x, flow, thresh=0.99999, mode="bilinear"):
    """
    warp an image/tensor (im2) back to im1, according to the optical flow

    x: [batch_size, channels, height, width] (im2)
    flow: [batch_size, 2, height, width] flow

    Returns (output * mask, mask) like the reference (imgflowarp.py:31-55).
    """
    if mode not in ("bilinear", "nearest"):
        raise ValueError(f"mode {mode} not in [bilinear, nearest]")
    return _WarpFunction.apply(x, flow, thresh, 0 if mode == "bilinear" else 1)


# the stacked pair loss hands the per-image maximum of its flow gradients to the raster backward (mr_render_flow_backward's
# grad_bound); False: that kernel finds its scale itself
PASS_GRADIENT_BOUND = True
# outputs="loss" on flows that carry their render's tile list (get_opticalflow(..., sparse_flows=True)): both pair kernels
# run over that list (mr_pair_consist_*_tiles) and the flow gradient is WRITTEN under the covered tiles only -- its one
# reader on the training path, mr_render_flow_backward, consults the same coverage bytes.  False: the dense kernels.
USE_TILE_LIST_KERNELS = True
# tests: fill the sparse gradient buffer with NaN first, so that a read outside the covered tiles shows up downstream
DEBUG_POISON_SPARSE_GRADS = False


class _PairConsistFunction(torch.autograd.Function):
    """Fused both-direction masked-L1 photometric loss.  Differentiable w.r.t. the two
    flows only (the images / jitter masks are data on the training path)."""

    @staticmethod
    def forward(ctx, flow12, flow21, image_ref, image, jitter_ref, jitter, thresh, want_debug, coverage=None, coverage_size=0,
                tiles=None):
        _lib.check_cuda(flow12, flow21, image_ref, image, jitter_ref, jitter)
        im_ref, im = _lib.contig(image_ref), _lib.contig(image)
        ctx.stacked = flow21 is None  # flow12 = [2B,H,W,2]: both flows in one tensor (get_opticalflow's fused path)
        if ctx.stacked:
            both = _lib.contig(flow12)
            if both.shape[0] != 2 * im.shape[0]:
                raise ValueError("stacked flows must be [2B, H, W, 2]")
            f12, f21 = both[: im.shape[0]], both[im.shape[0]:]
        else:
            f12, f21 = _lib.contig(flow12), _lib.contig(flow21)
        jm_ref, jm = _lib.contig(jitter_ref), _lib.contig(jitter)
        B, C, H, W = im.shape
        if C != 3 or im_ref.shape != im.shape:
            raise ValueError("images must be [B, 3, H, W]")
        if f12.shape != (B, H, W, 2) or f21.shape != (B, H, W, 2):
            raise ValueError("flows must be [B, H, W, 2]")
        Cj = jm.shape[1]
        if Cj not in (1, 3) or jm_ref.shape != jm.shape or jm.shape[2:] != (H, W):
            raise ValueError("jitter masks must be [B, 1 or 3, H, W]")
        dev = im.device
        # coverage bytes of the renders behind the flows ([2B, tiles_y, tiles_x, 4], opticalflow._StackedFlowFunction)
        hit12 = hit21 = None
        if coverage is not None:
            cs = int(coverage_size)
            if (coverage.dtype != torch.uint8 or not coverage.is_contiguous() or coverage.device != dev
                    or tuple(coverage.shape) != (2 * B, (cs + 7) // 8, (cs + 31) // 32, 4) or cs < H or cs < W):
                raise ValueError("coverage must be the [2B, tiles_y, tiles_x, 4] byte array of a raster of coverage_size")
            hit12, hit21 = coverage[:B], coverage[B:]
        ctx.coverage = (hit12, hit21, int(coverage_size))
        lib = _lib.load()
        sums = torch.empty((B, 4), dtype=torch.float32, device=dev)
        loss_fwd = torch.empty((B,), dtype=torch.float32, device=dev)
        loss_bwd = torch.empty((B,), dtype=torch.float32, device=dev)
        # the sparse contract: stacked flows with their render's tile list, nobody asking for per-pixel outputs
        listed = (tiles is not None and USE_TILE_LIST_KERNELS and ctx.stacked and hit12 is not None and not want_debug
                  and tiles[2] == coverage.numel() // 4)
        ctx.tiles = tiles if listed else None
        if listed:
            wbytes = int(lib.mr_pair_consist_tiles_workspace_bytes(B, int(coverage_size)))
            work = torch.empty((max(wbytes, 16),), dtype=torch.uint8, device=dev)
            _lib.call("mr_pair_consist_forward_tiles", _lib.ptr(f12), _lib.ptr(f21), _lib.ptr(im_ref), _lib.ptr(im),
                      _lib.ptr(jm_ref), _lib.ptr(jm), Cj, _lib.ptr(work), wbytes, _lib.ptr(sums), _lib.ptr(loss_fwd),
                      _lib.ptr(loss_bwd), B, H, W, float(thresh), _lib.ptr(hit12), _lib.ptr(hit21), int(coverage_size),
                      tiles[0], tiles[1], tiles[2], tiles[3], _lib.stream_ptr(dev))
            ctx.save_for_backward(f12, f21, im_ref, im, jm_ref, jm, sums)
            ctx.thresh = float(thresh)
            ctx.set_materialize_grads(False)
            return loss_fwd, loss_bwd
        wbytes = int(lib.mr_pair_consist_workspace_bytes(B, H, W))
        work = torch.empty((max(wbytes, 16),), dtype=torch.uint8, device=dev)
        dbg = [None] * 8
        if want_debug:
            fm1 = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
            fm2 = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
            dbg = [fm1, fm2] + [torch.empty((B, 3, H, W), dtype=torch.float32, device=dev) for _ in range(6)]
        _lib.call("mr_pair_consist_forward", _lib.ptr(f12), _lib.ptr(f21), _lib.ptr(im_ref), _lib.ptr(im),
                  _lib.ptr(jm_ref), _lib.ptr(jm), Cj, _lib.ptr(work), wbytes, _lib.ptr(sums),
                  _lib.ptr(loss_fwd), _lib.ptr(loss_bwd), *[_lib.ptr(t) for t in dbg], B, H, W, float(thresh),
                  _lib.ptr(hit12), _lib.ptr(hit21), int(coverage_size) if hit12 is not None else 0, _lib.stream_ptr(dev))
        ctx.save_for_backward(f12, f21, im_ref, im, jm_ref, jm, sums)
        ctx.thresh = float(thresh)
        ctx.set_materialize_grads(False)  # an unused direction arrives as None and is skipped
        outs = [loss_fwd, loss_bwd]
        if want_debug:
            ctx.mark_non_differentiable(*dbg)
            outs += dbg
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_fwd, g_bwd, *_):
        f12, f21, im_ref, im, jm_ref, jm, sums = ctx.saved_tensors
        B, _, H, W = im.shape
        if not (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) or (g_fwd is None and g_bwd is None):
            return (None,) * 11
        hit12, hit21, cov_size = ctx.coverage
        dev = im.device
        if g_fwd is None:
            g_fwd = torch.zeros((B,), dtype=torch.float32, device=dev)
        g_fwd = _lib.contig(g_fwd)
        g_bwd = _lib.contig(g_bwd) if g_bwd is not None else None
        gmax = None
        tiles = ctx.tiles
        if tiles is not None:
            # sparse contract: written under the covered tiles only (the raster backward reads nothing else)
            grad_both = (torch.full((2 * B, H, W, 2), float("nan"), dtype=torch.float32, device=dev) if DEBUG_POISON_SPARSE_GRADS
                         else torch.empty((2 * B, H, W, 2), dtype=torch.float32, device=dev))
            gmax = torch.zeros((2 * B,), dtype=torch.float32, device=dev) if PASS_GRADIENT_BOUND else None
            _lib.call("mr_pair_consist_backward_tiles", _lib.ptr(f12), _lib.ptr(f21), _lib.ptr(im_ref), _lib.ptr(im),
                      _lib.ptr(jm_ref), _lib.ptr(jm), int(jm.shape[1]), _lib.ptr(sums), _lib.ptr(g_fwd), _lib.ptr(g_bwd),
                      _lib.ptr(grad_both[:B]), _lib.ptr(grad_both[B:]), B, H, W, ctx.thresh, _lib.ptr(hit12), _lib.ptr(hit21),
                      cov_size, _lib.ptr(gmax), tiles[0], tiles[1], tiles[2], tiles[3], _lib.stream_ptr(dev))
            if gmax is not None:
                grad_both._hoc_grad_bound = (gmax, grad_both._version)
            return (grad_both,) + (None,) * 10
        if ctx.stacked:  # one gradient tensor for the stacked flows: no slice / cat nodes in autograd
            grad_both = torch.empty((2 * B, H, W, 2), dtype=torch.float32, device=dev)
            grad12, grad21 = grad_both[:B], grad_both[B:]
            # the kernel also leaves the largest |gradient| per image of the stack: the next consumer of these
            # gradients on the training path (the raster backward) needs it as its fixed-point scale and would
            # otherwise make a pass over its inputs to find it
            gmax = torch.zeros((2 * B,), dtype=torch.float32, device=dev) if PASS_GRADIENT_BOUND else None
        else:
            grad12 = torch.empty_like(f12)
            grad21 = torch.empty_like(f21)
        _lib.call("mr_pair_consist_backward", _lib.ptr(f12), _lib.ptr(f21), _lib.ptr(im_ref), _lib.ptr(im),
                  _lib.ptr(jm_ref), _lib.ptr(jm), int(jm.shape[1]), _lib.ptr(sums), _lib.ptr(g_fwd),
                  _lib.ptr(g_bwd), _lib.ptr(grad12), _lib.ptr(grad21), B, H, W, ctx.thresh, _lib.ptr(hit12), _lib.ptr(hit21),
                  cov_size if hit12 is not None else 0, _lib.ptr(gmax), _lib.stream_ptr(dev))
        if ctx.stacked:
            if gmax is not None:  # (rides on the gradient tensor, tied to its version like the coverage bytes of the flows)
                grad_both._hoc_grad_bound = (gmax, grad_both._version)
            return (grad_both,) + (None,) * 10
        return (grad12, grad21) + (None,) * 9


def _stacked_base(flow12, flow21):
    """The [2B,H,W,2] tensor the two flows are the halves of, if they are (get_opticalflow's fused path returns
    them that way): the pair function then differentiates that tensor directly and autograd needs neither
    slice nor cat nodes (each would copy a [2B]-sized gradient)."""
    base = getattr(flow12, "_base", None)
    if (base is None or base is not getattr(flow21, "_base", None) or not base.is_contiguous() or base.dim() != 4
            or flow12.shape != flow21.shape or base.shape[0] != 2 * flow12.shape[0] or base.shape[1:] != flow12.shape[1:]
            or not flow12.is_contiguous() or not flow21.is_contiguous() or flow12.data_ptr() != base.data_ptr()
            or flow21.data_ptr() != base.data_ptr() + flow12.numel() * flow12.element_size()):
        return None
    return base


def _is_fused_l1(criterion):
    """True for the reference's default criterion: PyramidCriterion('l1'), level_nb == 1."""
    return (
        getattr(criterion, "level_nb", None) == 1
        and isinstance(getattr(criterion, "criterion", None), torch.nn.L1Loss)
        and getattr(criterion.criterion, "reduction", None) == "none"
    )


# pair_consist(..., outputs=...): "full" returns masks / warps / diffs like the reference;
# "loss" skips materialising them (the trainer only consumes the loss)
DEFAULT_PAIR_OUTPUTS = "full"


def _coverage_of(stacked):
    """The coverage bytes ``get_opticalflow`` attached to its stacked flows -- (None, 0) when there are none or when
    the flows were written in place since (their version moved on: the bytes describe values that no longer exist)."""
    note = getattr(stacked, "_hoc_coverage", None) if stacked is not None else None
    if note is None or note[2] != stacked._version:
        return None, 0
    return note[0], note[1]


def _tiles_of(stacked):
    """... and the tile list of the renders behind them, when ``get_opticalflow(..., sparse_flows=True)`` produced flows
    that are defined under the covered tiles only (None otherwise)."""
    note = getattr(stacked, "_hoc_coverage", None) if stacked is not None else None
    if note is None or note[2] != stacked._version or len(note) < 4:
        return None
    # (a one-element marker: flows of opticalflow.flow_pair_loss's struct path, whose tile list lived in reusable scratch)
    return note[3] if (note[3] is not None and len(note[3]) >= 4) else None


def pair_consist(
    recons_flow,
    image_ref: torch.Tensor,
    image: torch.Tensor,
    jitter_mask_ref: torch.Tensor,
    jitter_mask: torch.Tensor,
    criterion,
    use_backward: bool = False,
    outputs=None,
):
    """
    We use the optical flow estimated at end frame which contains sampling offsets
    from start to end frame to warp the start image to the end one
    (reference imgflowarp.py:58-115).

    Args:
        recons_flow: [flow12, flow21], each [batch_size, height, width, 2]
        jitter_mask(_ref): locations that were outside of the original image before data
            augmentation
        image_ref: Image of reference (annotated) frame
        image: Image of unannotated frame
        outputs: "full" (reference behaviour) or "loss" (masks / warps / diffs are None)

    Returns:
        warp_loss [batch_size], masks, warps, diffs
    """
    outputs = outputs or DEFAULT_PAIR_OUTPUTS
    image_ref, image = image_ref.cuda(), image.cuda()
    jitter_mask_ref, jitter_mask = jitter_mask_ref.cuda(), jitter_mask.cuda()
    # (the fused kernels fetch the two taps of a row with one 8-byte load: images at least 2 wide)
    if _is_fused_l1(criterion) and image.shape[1] == 3 and jitter_mask.shape[1] in (1, 3) and image.shape[-1] >= 2:
        want_debug = outputs == "full"
        stacked = _stacked_base(recons_flow[0], recons_flow[1])
        coverage, coverage_size = _coverage_of(stacked)
        tiles = _tiles_of(stacked)
        if tiles is not None and (want_debug or not USE_TILE_LIST_KERNELS):
            raise ValueError('flows from get_opticalflow(..., sparse_flows=True) are defined under their renders\' coverage '
                             'only: pair_consist reads them with outputs="loss" (the tile-list kernels)')
        res = _PairConsistFunction.apply(stacked if stacked is not None else recons_flow[0],
                                         None if stacked is not None else recons_flow[1], image_ref, image,
                                         jitter_mask_ref, jitter_mask, 0.99999, want_debug, coverage, coverage_size, tiles)
        losses_fwd, losses_bwd = res[0], res[1]
        warp_loss = losses_bwd + losses_fwd if use_backward else losses_fwd
        if not want_debug:
            return warp_loss, None, None, None
        fm1, fm2, wm1, wm2, w1, w2, d1, d2 = res[2:]
        flow_mask1 = ~(recons_flow[1] == 0)
        flow_mask2 = ~(recons_flow[0] == 0)
        masks = [
            {"warp_mask": wm1, "full_mask": fm1.bool(), "flow_mask": flow_mask1},
            {"warp_mask": wm2, "full_mask": fm2.bool(), "flow_mask": flow_mask2},
        ]
        return warp_loss, masks, [w1, w2], [d1, d2]

    # generic criterion: the reference's composed structure on top of `warp`
    warp1, warp_mask1 = warp(image_ref, recons_flow[1].permute(0, 3, 1, 2))
    warpjitter1, _ = warp(jitter_mask_ref, recons_flow[0].permute(0, 3, 1, 2))
    warp2, warp_mask2 = warp(image, recons_flow[0].permute(0, 3, 1, 2))
    warpjitter2, _ = warp(jitter_mask, recons_flow[1].permute(0, 3, 1, 2))
    warp_mask1 = warp_mask1 * (warpjitter2 == 1).float()
    warp_mask2 = warp_mask2 * (warpjitter1 == 1).float()
    warps = [warp1, warp2]
    masks = []
    flow_mask1 = ~(recons_flow[1] == 0)
    valid_mask1 = warp_mask1[:, 0].bool() & flow_mask1[:, :, :, 0] & (jitter_mask[:, 0] == 1)
    masks.append({"warp_mask": warp_mask1, "full_mask": valid_mask1, "flow_mask": flow_mask1})
    flow_mask2 = ~(recons_flow[0] == 0)
    valid_mask2 = warp_mask2[:, 0].bool() & flow_mask2[:, :, :, 0] & (jitter_mask_ref[:, 0] == 1)
    masks.append({"warp_mask": warp_mask2, "full_mask": valid_mask2, "flow_mask": flow_mask2})
    _, _, losses_fwd, diffs_fwd, _ = criterion.compute(
        warp1, image, mask=valid_mask1.unsqueeze(1).repeat(1, 3, 1, 1)
    )
    _, _, losses_bwd, diffs_bwd, _ = criterion.compute(
        warp2, image_ref, mask=valid_mask2.unsqueeze(1).repeat(1, 3, 1, 1)
    )
    diffs = [diffs_fwd[0][:, :3], diffs_bwd[0][:, :3]]
    warp_loss = losses_bwd + losses_fwd if use_backward else losses_fwd
    return warp_loss, masks, warps, diffs


def get_occlusion_mask(mask_flow1, mask_flow2, flow12, flow21):
    """
    Perform forward-backward consistency check by warping a grid which contains the pixel
    locations from frame1 to frame2 and back with the optical flows; pixels whose round
    trip moves by more than 0.03 (normalised units) are occluded in one of the views
    (reference imgflowarp.py:118-146).

    mask_flow*: [B, 1, H, W]; flow*: [B, >=2, H, W] (first two channels used).
    Returns occl_mask1, occl_mask2: [B, H, W].
    """
    _lib.check_cuda(mask_flow1, mask_flow2, flow12, flow21)
    m1, m2 = _lib.contig(mask_flow1.detach()), _lib.contig(mask_flow2.detach())
    f12, f21 = _lib.contig(flow12.detach()), _lib.contig(flow21.detach())
    B, _, H, W = m1.shape
    if f12.shape[0] != B or f12.shape[2:] != (H, W) or f12.shape != f21.shape or f12.shape[1] < 2:
        raise ValueError("flows must be [B, >=2, H, W] matching the masks")
    occl1 = torch.empty((B, H, W), dtype=torch.float32, device=m1.device)
    occl2 = torch.empty((B, H, W), dtype=torch.float32, device=m1.device)
    _lib.call("mr_occlusion_mask", _lib.ptr(m1), _lib.ptr(m2), _lib.ptr(f12), _lib.ptr(f21),
              int(f12.shape[1]) * H * W, None, None, _lib.ptr(occl1), _lib.ptr(occl2), B, H, W, 0.03, 0.99999,
              _lib.stream_ptr(m1.device))
    return occl1, occl2


def occlusion_mask_from_warped_grid(
    grid: torch.Tensor, warped_grid: torch.Tensor, distance_thresh=0.03
) -> torch.Tensor:
    """
    Args:
        grid: (batch_size, height, width, >=3): x, y locations then the mask
        warped_grid: same layout, after the forward then backward warps

    Returns:
        valid_mask: locations which pass the forward-backward consistency check
        (reference imgflowarp.py:149-172; plain tensor ops, used stand-alone only)
    """
    mask = grid[:, :, :, 2] * warped_grid[:, :, :, 2]
    grid_displs = ((warped_grid - grid) * mask.unsqueeze(-1))[:, :, :, :2].norm(2, -1)
    motion_mask = (grid_displs < distance_thresh).float()
    valid_mask = mask * motion_mask
    return valid_mask
