"""BatchNorm2d with frozen statistics fused with the residual add and ReLU that follow it
(``mr_bn_act_forward`` / ``mr_bn_act_backward``).  The reference trains with ``--freeze_batchnorm``
(trainmeshwarp.py:205-206, 237-240): BatchNorm layers in eval mode, affine parameters trainable, so
``relu(bn(x))`` / ``relu(bn(x) + identity)`` / ``bn(x)`` of resnet.py:46-58 are per-channel affine maps."""
import torch

from handobjectconsist_amd import _lib


_ACT_DTYPES = {torch.float32: 0, torch.bfloat16: 1}  # act_dtype of the C-ABI


def _nhwc_channels_ok(C):
    return 4 <= C <= 1024 and 1024 % C == 0


def _layout(x):
    """(tensor in a layout the kernels take, channels_last flag): channels-last 4-D activations (what the trunk runs
    in, MIOpen's convolutions are faster there) stay as they are; everything else is made NCHW-contiguous."""
    if (x.dim() == 4 and _nhwc_channels_ok(x.shape[1]) and not x.is_contiguous()
            and x.is_contiguous(memory_format=torch.channels_last)):
        return x, 1
    return x.contiguous(), 0


class _BnActFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, weight, bias, running_mean, running_var, eps, relu, dup):
        ctx.set_materialize_grads(False)
        _lib.check_cuda(x, residual, weight, bias, running_mean, running_var)
        if x.dim() < 2 or x.dtype not in _ACT_DTYPES:
            raise ValueError("expected an fp32 or bf16 [N, C, ...] tensor")
        xc, cl = _layout(x)
        rc = None
        if residual is not None:
            if residual.shape != xc.shape or residual.dtype != xc.dtype:
                raise ValueError("residual must match x")
            rc = residual.contiguous(memory_format=torch.channels_last) if cl else residual.contiguous()
        N, C = xc.shape[:2]
        plane = xc[0, 0].numel() if N and C else 0
        w, b = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        m, v = running_mean.float().contiguous(), running_var.float().contiguous()
        if not (w.shape == b.shape == m.shape == v.shape == (C,)):
            raise ValueError("channel arrays must be [C]")
        y = torch.empty_like(xc)
        _lib.call("mr_bn_act_forward", _lib.ptr(xc), _lib.ptr(rc), _lib.ptr(w), _lib.ptr(b), _lib.ptr(m), _lib.ptr(v),
                  float(eps), int(bool(relu)), _ACT_DTYPES[xc.dtype], cl, _lib.ptr(y), N, C, plane,
                  _lib.stream_ptr(xc.device))
        ctx.save_for_backward(xc, rc, w, b, m, v)
        ctx.cfg = (float(eps), bool(relu), N, C, plane, cl)
        # dup: the same activation as two autograd outputs (one per consumer); their gradients then arrive
        # separately and are summed inside the backward kernel instead of by a separate add pass
        return (y, y.view_as(y)) if dup else y

    @staticmethod
    def backward(ctx, grad_y, grad_y2=None):
        xc, rc, w, b, m, v = ctx.saved_tensors
        eps, relu, N, C, plane, cl = ctx.cfg
        need_x, need_r, need_w, need_b = ctx.needs_input_grad[:4]
        if grad_y is None:
            grad_y, grad_y2 = grad_y2, None
        if grad_y is None:
            return (None,) * 9
        fmt = torch.channels_last if cl else torch.contiguous_format
        g = grad_y.to(xc.dtype).contiguous(memory_format=fmt)
        g2 = grad_y2.to(xc.dtype).contiguous(memory_format=fmt) if grad_y2 is not None else None
        dev = xc.device
        grad_x = torch.empty_like(xc)
        grad_r = torch.empty_like(xc) if (rc is not None and need_r) else None
        grad_w = torch.empty_like(w) if need_w else None
        grad_b = torch.empty_like(b) if need_b else None
        wbytes = int(_lib.load().mr_bn_act_backward_workspace_bytes(N, C))
        work = torch.empty((wbytes,), dtype=torch.uint8, device=dev) if (need_w or need_b) else None
        _lib.call("mr_bn_act_backward", _lib.ptr(g), _lib.ptr(g2), _lib.ptr(xc), _lib.ptr(rc), _lib.ptr(w), _lib.ptr(b), _lib.ptr(m),
                  _lib.ptr(v), eps, int(relu), _ACT_DTYPES[xc.dtype], cl, _lib.ptr(grad_x), _lib.ptr(grad_r), _lib.ptr(grad_w),
                  _lib.ptr(grad_b),
                  _lib.ptr(work), wbytes, N, C, plane, _lib.stream_ptr(dev))
        return (grad_x if need_x else None), grad_r, grad_w, grad_b, None, None, None, None, None


def bn_act(x, bn, residual=None, relu=True, dup=False):
    """``relu(bn(x) [+ residual])`` for an ``nn.BatchNorm2d`` in eval mode (running statistics), one kernel.
    ``dup=True`` returns the result twice (two autograd outputs over one buffer) for an activation with two
    consumers -- a block's convolution and its identity branch -- whose gradients the backward kernel then sums."""
    if bn.training or not bn.track_running_stats:
        raise RuntimeError("bn_act needs frozen BatchNorm statistics (module.eval())")
    return _BnActFunction.apply(x, residual, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, relu, dup)


class _StemPoolFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, dup):
        ctx.set_materialize_grads(False)
        _lib.check_cuda(x, weight, bias, running_mean, running_var)
        if x.dim() != 4 or x.dtype not in _ACT_DTYPES:
            raise ValueError("expected an fp32 or bf16 [N, C, H, W] tensor")
        xc, cl = _layout(x)
        N, C, H, W = xc.shape
        w, b = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        m, v = running_mean.float().contiguous(), running_var.float().contiguous()
        oshape = (N, C, (H - 1) // 2 + 1 if H else 0, (W - 1) // 2 + 1 if W else 0)
        fmt = torch.channels_last if cl else torch.contiguous_format
        y = torch.empty(oshape, dtype=xc.dtype, device=xc.device, memory_format=fmt)
        # channels-last: the forward keeps every pooled value's arg-max position (1 byte) for the backward
        argmax = torch.empty(oshape, dtype=torch.uint8, device=xc.device, memory_format=fmt) if cl else None
        _lib.call("mr_stem_pool_forward", _lib.ptr(xc), _lib.ptr(w), _lib.ptr(b), _lib.ptr(m), _lib.ptr(v), float(eps),
                  _ACT_DTYPES[xc.dtype], cl, _lib.ptr(y), _lib.ptr(argmax), N, C, H, W, _lib.stream_ptr(xc.device))
        ctx.save_for_backward(xc, w, b, m, v, argmax)
        ctx.cfg = (float(eps), cl)
        return (y, y.view_as(y)) if dup else y

    @staticmethod
    def backward(ctx, grad_y, grad_y2=None):
        xc, w, b, m, v, argmax = ctx.saved_tensors
        eps, cl = ctx.cfg
        N, C, H, W = xc.shape
        need_x, need_w, need_b = ctx.needs_input_grad[:3]
        if grad_y is None:
            grad_y, grad_y2 = grad_y2, None
        if grad_y is None:
            return (None,) * 7
        fmt = torch.channels_last if cl else torch.contiguous_format
        g = grad_y.to(xc.dtype).contiguous(memory_format=fmt)
        g2 = grad_y2.to(xc.dtype).contiguous(memory_format=fmt) if grad_y2 is not None else None
        grad_x = torch.empty_like(xc)
        grad_w = torch.empty_like(w) if need_w else None
        grad_b = torch.empty_like(b) if need_b else None
        wbytes = int(_lib.load().mr_stem_pool_backward_workspace_bytes(N, C, H, W))
        work = torch.empty((wbytes,), dtype=torch.uint8, device=xc.device) if (need_w or need_b) else None
        _lib.call("mr_stem_pool_backward", _lib.ptr(g), _lib.ptr(g2), _lib.ptr(xc), _lib.ptr(argmax), _lib.ptr(w), _lib.ptr(b),
                  _lib.ptr(m),
                  _lib.ptr(v), eps, _ACT_DTYPES[xc.dtype], cl, _lib.ptr(grad_x), _lib.ptr(grad_w), _lib.ptr(grad_b),
                  _lib.ptr(work), wbytes, N, C, H, W,
                  _lib.stream_ptr(xc.device))
        return (grad_x if need_x else None), grad_w, grad_b, None, None, None, None


def stem_pool(x, bn, dup=False):
    """``MaxPool2d(3, 2, 1)(relu(bn(x)))`` for an ``nn.BatchNorm2d`` in eval mode: the ResNet stem, one kernel
    (``dup``: see ``bn_act``)."""
    if bn.training or not bn.track_running_stats:
        raise RuntimeError("stem_pool needs frozen BatchNorm statistics (module.eval())")
    return _StemPoolFunction.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, dup)
