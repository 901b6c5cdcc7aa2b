"""The frame-pair step of the training path through ONE C call each way (ABI 8: ``mr_pair_step_forward`` /
``mr_pair_step_backward``, include/meshraster_hip.h) -- the host side of ``opticalflow.flow_pair_loss`` for meshes handed over
as (hand, object) parts when the vertices want a gradient.

What it replaces on the host (same kernels, same values): two autograd nodes, five ctypes calls of 30 - 60 scalars each, ~20
``torch.empty`` per pass, the pair's ``mean`` and its backward (warpbranch.py:87-88) as PyTorch launches.  Here a PLAN per
(shape, device, stream, renderer settings) owns a ``MrPairStep`` struct whose size fields are filled once, the scratch buffer
the forward's launches hand to one another (nothing reads it after they ran: one per plan serves every call on its stream),
and the pinned word the render reports its tile-list length to; a call allocates three tensors (what the backward reads,
the flows, the losses), writes ~25 pointers and makes one call.

Reference semantics: /root/reference/meshreg/warping/opticalflow.py:51-156 (detach_textures=False, detach_renders=True) +
meshreg/warping/imgflowarp.py:58-115 + meshreg/optim/pyramidloss.py:56-62 + meshreg/optim/lossutils.py:1-8 for one pair, and
the batch mean of meshreg/models/warpbranch.py:87-88.
"""
import ctypes

import torch

from handobjectconsist_amd import _lib
from handobjectconsist_amd.utils import textutils

_c = ctypes
_I32, _F32, _PTR, _I64 = _c.c_int32, _c.c_float, _c.c_void_p, _c.c_int64

# field order = declaration order of MrPairStep (checked against mr_pair_step_field_offsets by tests/test_abi.py)
SIZE_FIELDS = ("batch_size", "num_verts_a", "num_verts_b", "num_hand_faces", "num_obj_faces", "hand_faces_batched", "fill_back",
               "image_size", "height", "width", "jitter_channels", "cam_batched", "n_lut", "bg_stride", "texel_layout", "want_grad",
               "mean_of", "flags")
FLOAT_FIELDS = ("orig_size", "near_", "far_", "eps", "alpha_thresh", "distance_thresh", "warp_thresh", "pair_thresh")
INPUT_FIELDS = ("verts1a", "verts1b", "verts2a", "verts2b", "K1", "K2", "R", "t", "dist_coeffs", "hand_faces", "obj_faces",
                "keep_lut", "background", "image_ref", "image", "jitter_ref", "jitter")
GRAD_FIELDS = ("grad_loss_fwd", "grad_loss_bwd", "grad_loss_sum", "grad_mean", "grad_verts1a", "grad_verts1b", "grad_verts2a",
               "grad_verts2b")
GRAD_BUFFER_USED = 1  # MR_PAIR_STEP_GRAD_BUFFER_USED
SEPARATE_LAUNCHES = 2  # MR_PAIR_STEP_SEPARATE_LAUNCHES: every stage a launch of its own (tests, profiling)
LIST_CLEAN = 4  # MR_PAIR_STEP_LIST_CLEAN: the scratch's tile-list header was left clean by the previous forward call on it


class MrPairStep(ctypes.Structure):
    _fields_ = ([(n, _I32) for n in SIZE_FIELDS] + [(n, _F32) for n in FLOAT_FIELDS] + [(n, _PTR) for n in INPUT_FIELDS]
                + [("scratch", _PTR), ("saved", _PTR), ("scratch_bytes", _I64), ("saved_bytes", _I64), ("flows", _PTR),
                   ("losses", _PTR), ("tile_count_out", _PTR), ("tile_bound", _I64)] + [(n, _PTR) for n in GRAD_FIELDS])


_PLANS = {}
_FACES64 = {}
_COUNT_WORDS = {}


def _faces64(t, first_only=False):
    """contiguous int64 form of a face tensor (``first_only``: of its first batch entry), kept while the source tensor is the
    same object at the same version"""
    key = (id(t), first_only)
    hit = _FACES64.get(key)
    if hit is None or hit[0] is not t or hit[1] != t._version:
        if len(_FACES64) > 64:
            _FACES64.clear()
        src = t[0] if first_only else t
        c = src if (src.dtype == torch.int64 and src.is_contiguous()) else src.to(torch.int64).contiguous()
        hit = (t, t._version, c)
        _FACES64[key] = hit
    return hit[2]


class _Plan:
    """Everything about a pair step that does not change from call to call."""

    def __init__(self, dev, stream, sizes, floats, mean_of):
        lib = _lib.load()
        self.dev, self.stream = dev, _c.c_void_p(stream)
        self.fwd, self.bwd = lib.mr_pair_step_forward, lib.mr_pair_step_backward
        self.st, self.st_b = MrPairStep(), MrPairStep()
        for st in (self.st, self.st_b):
            for n, v in zip(SIZE_FIELDS, sizes):
                setattr(st, n, int(v))
            for n, v in zip(FLOAT_FIELDS, floats):
                setattr(st, n, float(v))
            st.mean_of = int(mean_of)
        sc, sv, th = _I64(), _I64(), _I64()
        rc = lib.mr_pair_step_sizes(_c.byref(self.st), _c.byref(sc), _c.byref(sv), _c.byref(th))
        self.ok = rc == 0
        if rc not in (0, -2):
            raise RuntimeError(f"mr_pair_step_sizes failed: {rc}")
        self.scratch_bytes, self.saved_bytes, self.tile_hit_offset = int(sc.value), int(sv.value), int(th.value)
        self.scratch = None
        # the render's tile-list header inside the scratch is re-zeroed by the LAST launch of a forward call: the next call on
        # this plan says so (LIST_CLEAN) and goes without a clearing launch.  False: never used, poisoned, or a call failed.
        self.list_clean = False
        self.base_flags = int(self.st.flags) & ~(GRAD_BUFFER_USED | LIST_CLEAN)
        self.st_addr, self.st_b_addr = _c.c_void_p(_c.addressof(self.st)), _c.c_void_p(_c.addressof(self.st_b))
        B, is_ = self.st.batch_size, self.st.image_size
        self.B, self.H, self.W, self.is_ = B, self.st.height, self.st.width, is_
        self.tile_hit_shape = (2 * B, (is_ + 7) // 8, (is_ + 31) // 32, 4)
        self.tile_hit_bytes = 2 * B * ((is_ + 7) // 8) * ((is_ + 31) // 32) * 4
        # the render's tile-list length of the previous call = the guess for the next: a pinned host word the kernel writes,
        # shared by the plans of a (device, batch, raster) whatever their stream (a guess: any value gives the same images).
        # A plan first built while its stream is being captured into a graph finds the word of the eager passes before it, or
        # goes without (pinning memory is not a capturable operation): the render then dispatches a quarter of the tiles.
        wkey = (dev.index, B, is_)
        word = _COUNT_WORDS.get(wkey)
        if word is None and not torch.cuda.is_current_stream_capturing():
            word = torch.zeros(1, dtype=torch.int32).pin_memory()
            _COUNT_WORDS[wkey] = word
        self.count_word = word
        self.count = _c.c_uint32.from_address(word.data_ptr()) if word is not None else None
        self.const = None  # (sources, device copies) of R / t / dist_coeffs / background / lut

    def ensure_scratch(self):
        if self.scratch is None:
            self.scratch = torch.empty((max(self.scratch_bytes, 16),), dtype=torch.uint8, device=self.dev)
            self.st.scratch = self.scratch.data_ptr()
            self.st.scratch_bytes = self.scratch_bytes
        return self.scratch


def _f32c(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.to(torch.float32).contiguous()


class _PairStepFunction(torch.autograd.Function):
    """(hand1, obj1, hand2, obj2 [B,V*,3]) -> (mean, loss_sum[B], loss_fwd[B], loss_bwd[B], flows[2B,H,W,2]); everything else
    rides in ``call`` (plan, non-differentiable tensors).  Differentiable w.r.t. the four vertex tensors."""

    @staticmethod
    def forward(ctx, h1, o1, h2, o2, call):
        ctx.set_materialize_grads(False)
        plan, K1, K2, consts, hf, of, images, poison, out = call
        dev, B = plan.dev, plan.B
        st = plan.st
        saved = torch.empty((plan.saved_bytes,), dtype=torch.uint8, device=dev)
        flows = torch.empty((2 * B, plan.H, plan.W, 2), dtype=torch.float32, device=dev)
        losses = torch.empty((4 * B + 1,), dtype=torch.float32, device=dev)  # (fwd | bwd | sum | mean | B words of scratch)
        scratch = plan.ensure_scratch()
        if poison:  # (tests: whatever a sparse output is not supposed to be read from holds NaN / -1 / INT_MIN face indices)
            saved.fill_(255)
            saved[:2 * B * plan.is_ * plan.is_ * 4].view(torch.int32).fill_(-2 ** 31)  # (the face index map leads the buffer)
            scratch.fill_(255)
            flows.fill_(float("nan"))
            plan.list_clean = False
        st.verts1a, st.verts1b, st.verts2a, st.verts2b = h1.data_ptr(), o1.data_ptr(), h2.data_ptr(), o2.data_ptr()
        st.K1, st.K2 = K1.data_ptr(), K2.data_ptr()
        R, t, dist, bg, lut = consts
        st.R, st.t, st.dist_coeffs, st.background = R.data_ptr(), t.data_ptr(), dist.data_ptr(), bg.data_ptr()
        st.keep_lut = lut.data_ptr() if lut is not None else None
        st.hand_faces, st.obj_faces = hf.data_ptr(), of.data_ptr()
        im_ref, im, jm_ref, jm = images
        st.image_ref, st.image, st.jitter_ref, st.jitter = im_ref.data_ptr(), im.data_ptr(), jm_ref.data_ptr(), jm.data_ptr()
        st.saved, st.saved_bytes = saved.data_ptr(), plan.saved_bytes
        st.flows, st.losses = flows.data_ptr(), losses.data_ptr()
        last = plan.count.value if plan.count is not None else 0  # host memory: no device synchronisation
        st.tile_bound = (last + last // 8 + 64) if last > 0 else -1
        st.tile_count_out = plan.count_word.data_ptr() if plan.count_word is not None else None
        want = any(ctx.needs_input_grad[:4])
        st.want_grad = 1 if want else 0
        st.flags = plan.base_flags | (LIST_CLEAN if plan.list_clean else 0)
        plan.list_clean = False  # (until this call has returned MR_OK)
        if dev.index != torch.cuda.current_device():
            with torch.cuda.device(dev):
                rc = plan.fwd(plan.st_addr, plan.stream)
        else:
            rc = plan.fwd(plan.st_addr, plan.stream)
        if rc != 0:
            raise RuntimeError(f"mr_pair_step_forward failed: {rc}")
        plan.list_clean = True
        ctx.plan, ctx.used = plan, False
        if want:
            ctx.save_for_backward(saved, h1, o1, h2, o2, K1, K2)
        ctx.mark_non_differentiable(flows)
        out.append(saved)  # (the coverage words of the flows live in it: pair_step makes the view)
        return losses[3 * B], losses[2 * B:3 * B], losses[:B], losses[B:2 * B], flows

    @staticmethod
    def backward(ctx, g_mean, g_sum, g_fwd, g_bwd, _g_flows):
        if not any(ctx.needs_input_grad[:4]) or (g_mean is None and g_sum is None and g_fwd is None and g_bwd is None):
            return (None,) * 5
        plan = ctx.plan
        saved, h1, o1, h2, o2, K1, K2 = ctx.saved_tensors
        st = plan.st_b
        st.saved, st.saved_bytes = saved.data_ptr(), plan.saved_bytes
        st.verts1a, st.verts1b, st.verts2a, st.verts2b = h1.data_ptr(), o1.data_ptr(), h2.data_ptr(), o2.data_ptr()
        st.K1, st.K2 = K1.data_ptr(), K2.data_ptr()
        keep = [_f32c(g) if g is not None else None for g in (g_fwd, g_bwd, g_sum, g_mean)]
        st.grad_loss_fwd, st.grad_loss_bwd, st.grad_loss_sum, st.grad_mean = [g.data_ptr() if g is not None else None for g in keep]
        grads = [torch.empty_like(x) if w else None for x, w in zip((h1, o1, h2, o2), ctx.needs_input_grad[:4])]
        st.grad_verts1a, st.grad_verts1b, st.grad_verts2a, st.grad_verts2b = [g.data_ptr() if g is not None else None for g in grads]
        st.want_grad = 1
        # (a second backward through this node: the buffer is cleared first; the plan's own bits -- SEPARATE_LAUNCHES -- stay)
        st.flags = plan.base_flags | (GRAD_BUFFER_USED if ctx.used else 0)
        ctx.used = True
        if plan.dev.index != torch.cuda.current_device():
            with torch.cuda.device(plan.dev):
                rc = plan.bwd(plan.st_b_addr, plan.stream)
        else:
            rc = plan.bwd(plan.st_b_addr, plan.stream)
        if rc != 0:
            raise RuntimeError(f"mr_pair_step_backward failed: {rc}")
        return grads[0], grads[1], grads[2], grads[3], None


def pair_step(parts1, parts2, hand_face, obj_faces, K1, K2, neurenderer, is_, H, W, image_ref, image, jitter_ref, jitter, lut,
              mean_of_fwd_only=False, poison=False, flags=0):
    """One frame pair through the two struct calls.  Returns ``(mean, loss_sum, loss_fwd, loss_bwd, flows[2B,H,W,2], tile_hit)``
    or None where the fused path does not apply (sizes: mr_pair_step_sizes says MR_ERR_NOTIMPL).  Callers
    (``opticalflow.flow_pair_loss``) have checked devices / dtypes / shapes of the tensors they pass."""
    h1, o1 = parts1
    h2, o2 = parts2
    dev = h1.device
    B, Va, Vb = h1.shape[0], h1.shape[1], o1.shape[1]
    hand_batched = hand_face.dim() == 3 and hand_face.shape[0] == B and B > 1
    hf = _faces64(hand_face, first_only=not (hand_face.dim() == 2 or hand_batched))
    of = _faces64(obj_faces)
    Fh, Fo, Cj = hf.shape[-2], of.shape[1], jitter.shape[1]
    stream = torch._C._cuda_getCurrentRawStream(dev.index)  # (what torch.cuda.current_stream(dev).cuda_stream returns, without the object)
    R_src, t_src, d_src, bg_src = neurenderer.R, neurenderer.t, neurenderer.dist_coeffs, neurenderer.background_color
    nb = R_src.shape[0] if R_src.dim() == 3 else 1
    floats = (float(neurenderer.orig_size), float(neurenderer.near), float(neurenderer.far), float(neurenderer.rasterizer_eps),
              0.99999, 0.03, 0.99999, 0.99999)
    key = (dev.index, stream, B, Va, Vb, Fh, Fo, hand_batched, is_, H, W, Cj, bool(neurenderer.fill_back), nb, floats,
           0 if lut is None else lut.numel(), bool(mean_of_fwd_only), int(flags))
    plan = _PLANS.get(key)
    if plan is None:
        from handobjectconsist_amd.neurender import rasterize

        bg, bg_stride = rasterize._background_tensor(bg_src, dev, 2 * B)
        sizes = (B, Va, Vb, Fh, Fo, int(hand_batched), int(bool(neurenderer.fill_back)), is_, H, W, Cj, int(nb == B and B > 1),
                 0 if lut is None else lut.numel(), bg_stride, textutils.texel_layout_code(), 0, int(mean_of_fwd_only), int(flags))
        plan = _Plan(dev, stream, sizes, floats, mean_of_fwd_only)
        if len(_PLANS) > 32:
            _PLANS.clear()
        _PLANS[key] = plan
    if not plan.ok:
        return None
    c = plan.const
    if (c is None or c[0] is not R_src or c[1] is not t_src or c[2] is not d_src or c[3] is not bg_src or c[4] is not lut
            or c[5] != (R_src._version, t_src._version, d_src._version)):
        from handobjectconsist_amd.neurender import rasterize

        bg, _ = rasterize._background_tensor(bg_src, dev, 2 * B)
        Rc = _f32c(R_src.detach().to(dev).reshape(-1, 3, 3))
        tc = _f32c(t_src.detach().to(dev).reshape(-1, 3))
        dc = _f32c(d_src.detach().to(dev).reshape(-1, 5))
        if Rc.shape[0] not in (1, B) or tc.shape[0] != Rc.shape[0] or dc.shape[0] != Rc.shape[0]:
            raise ValueError("expected R / t / dist_coeffs with batch 1 or B")
        c = (R_src, t_src, d_src, bg_src, lut, (R_src._version, t_src._version, d_src._version), (Rc, tc, dc, bg, lut))
        plan.const = c
    if plan.st.texel_layout != textutils.texel_layout_code():  # (the texel table is a module switch: follow it)
        plan.st.texel_layout = plan.st_b.texel_layout = textutils.texel_layout_code()
    K1c, K2c = _f32c(K1), _f32c(K2)
    if K1c.shape != (B, 3, 3) or K2c.shape != (B, 3, 3):
        raise ValueError("expected intrinsics [B,3,3]")
    images = (_f32c(image_ref), _f32c(image), _f32c(jitter_ref), _f32c(jitter))
    if (images[1].shape != (B, 3, H, W) or images[0].shape != images[1].shape or images[3].shape != (B, Cj, H, W)
            or images[2].shape != images[3].shape):
        raise ValueError("images must be [B,3,H,W] and jitter masks [B,1 or 3,H,W] of the flows' size")
    out = []
    call = (plan, K1c, K2c, c[6], hf, of, images, poison, out)
    tens = [x if x.is_contiguous() else x.contiguous() for x in (h1, o1, h2, o2)]
    mean, loss_sum, loss_fwd, loss_bwd, flows = _PairStepFunction.apply(tens[0], tens[1], tens[2], tens[3], call)
    saved = out[0]
    tile_hit = saved[plan.tile_hit_offset:plan.tile_hit_offset + plan.tile_hit_bytes].view(plan.tile_hit_shape)
    return mean, loss_sum, loss_fwd, loss_bwd, flows, tile_hit
