"""BatchNorm2d with frozen statistics fused with the residual add and ReLU that follow it
(``mr_bn_act_forward`` / ``mr_bn_act_backward``).  The reference trains with ``--freeze_batchnorm``
(trainmeshwarp.py:205-206, 237-240): BatchNorm layers in eval mode, affine parameters trainable, so
``relu(bn(x))`` / ``relu(bn(x) + identity)`` / ``bn(x)`` of resnet.py:46-58 are per-channel affine maps."""
import torch

from handobjectconsist_amd import _lib


_ACT_DTYPES = {torch.float32: 0, torch.bfloat16: 1}  # act_dtype of the C-ABI


class _BnActFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, weight, bias, running_mean, running_var, eps, relu):
        _lib.check_cuda(x, residual, weight, bias, running_mean, running_var)
        if x.dim() < 2 or x.dtype not in _ACT_DTYPES:
            raise ValueError("expected an fp32 or bf16 [N, C, ...] tensor")
        xc = x.contiguous()
        rc = residual.contiguous() if residual is not None else None
        if rc is not None and (rc.shape != xc.shape or rc.dtype != xc.dtype):
            raise ValueError("residual must match x")
        N, C = xc.shape[:2]
        plane = xc[0, 0].numel() if N and C else 0
        w, b = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        m, v = running_mean.float().contiguous(), running_var.float().contiguous()
        if not (w.shape == b.shape == m.shape == v.shape == (C,)):
            raise ValueError("channel arrays must be [C]")
        y = torch.empty_like(xc)
        _lib.call("mr_bn_act_forward", _lib.ptr(xc), _lib.ptr(rc), _lib.ptr(w), _lib.ptr(b), _lib.ptr(m), _lib.ptr(v),
                  float(eps), int(bool(relu)), _ACT_DTYPES[xc.dtype], _lib.ptr(y), N, C, plane, _lib.stream_ptr(xc.device))
        ctx.save_for_backward(xc, rc, w, b, m, v)
        ctx.cfg = (float(eps), bool(relu), N, C, plane)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        xc, rc, w, b, m, v = ctx.saved_tensors
        eps, relu, N, C, plane = ctx.cfg
        need_x, need_r, need_w, need_b = ctx.needs_input_grad[:4]
        g = grad_y.to(xc.dtype).contiguous()
        dev = xc.device
        grad_x = torch.empty_like(xc)
        grad_r = torch.empty_like(xc) if (rc is not None and need_r) else None
        grad_w = torch.empty_like(w) if need_w else None
        grad_b = torch.empty_like(b) if need_b else None
        wbytes = int(_lib.load().mr_bn_act_backward_workspace_bytes(N, C))
        work = torch.empty((wbytes,), dtype=torch.uint8, device=dev) if (need_w or need_b) else None
        _lib.call("mr_bn_act_backward", _lib.ptr(g), _lib.ptr(xc), _lib.ptr(rc), _lib.ptr(w), _lib.ptr(b), _lib.ptr(m),
                  _lib.ptr(v), eps, int(relu), _ACT_DTYPES[xc.dtype], _lib.ptr(grad_x), _lib.ptr(grad_r), _lib.ptr(grad_w),
                  _lib.ptr(grad_b),
                  _lib.ptr(work), wbytes, N, C, plane, _lib.stream_ptr(dev))
        return (grad_x if need_x else None), grad_r, grad_w, grad_b, None, None, None, None


def bn_act(x, bn, residual=None, relu=True):
    """``relu(bn(x) [+ residual])`` for an ``nn.BatchNorm2d`` in eval mode (running statistics), one kernel."""
    if bn.training or not bn.track_running_stats:
        raise RuntimeError("bn_act needs frozen BatchNorm statistics (module.eval())")
    return _BnActFunction.apply(x, residual, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, relu)


class _StemPoolFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps):
        _lib.check_cuda(x, weight, bias, running_mean, running_var)
        if x.dim() != 4 or x.dtype not in _ACT_DTYPES:
            raise ValueError("expected an fp32 or bf16 [N, C, H, W] tensor")
        xc = x.contiguous()
        N, C, H, W = xc.shape
        w, b = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        m, v = running_mean.float().contiguous(), running_var.float().contiguous()
        y = torch.empty((N, C, (H - 1) // 2 + 1 if H else 0, (W - 1) // 2 + 1 if W else 0), dtype=xc.dtype,
                        device=xc.device)
        _lib.call("mr_stem_pool_forward", _lib.ptr(xc), _lib.ptr(w), _lib.ptr(b), _lib.ptr(m), _lib.ptr(v), float(eps),
                  _ACT_DTYPES[xc.dtype], _lib.ptr(y), N, C, H, W, _lib.stream_ptr(xc.device))
        ctx.save_for_backward(xc, w, b, m, v)
        ctx.eps = float(eps)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        xc, w, b, m, v = ctx.saved_tensors
        N, C, H, W = xc.shape
        need_x, need_w, need_b = ctx.needs_input_grad[:3]
        g = grad_y.to(xc.dtype).contiguous()
        grad_x = torch.empty_like(xc)
        grad_w = torch.empty_like(w) if need_w else None
        grad_b = torch.empty_like(b) if need_b else None
        wbytes = int(_lib.load().mr_stem_pool_backward_workspace_bytes(N, C, H, W))
        work = torch.empty((wbytes,), dtype=torch.uint8, device=xc.device) if (need_w or need_b) else None
        _lib.call("mr_stem_pool_backward", _lib.ptr(g), _lib.ptr(xc), _lib.ptr(w), _lib.ptr(b), _lib.ptr(m), _lib.ptr(v),
                  ctx.eps, _ACT_DTYPES[xc.dtype], _lib.ptr(grad_x), _lib.ptr(grad_w), _lib.ptr(grad_b), _lib.ptr(work), wbytes,
                  N, C, H, W,
                  _lib.stream_ptr(xc.device))
        return (grad_x if need_x else None), grad_w, grad_b, None, None, None


def stem_pool(x, bn):
    """``MaxPool2d(3, 2, 1)(relu(bn(x)))`` for an ``nn.BatchNorm2d`` in eval mode: the ResNet stem, one kernel."""
    if bn.training or not bn.track_running_stats:
        raise RuntimeError("stem_pool needs frozen BatchNorm statistics (module.eval())")
    return _StemPoolFunction.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
