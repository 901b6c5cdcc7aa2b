"""Fused frozen-BatchNorm + residual + ReLU (mr_bn_act_*) against the stock PyTorch modules it replaces in the
ResNet-18 trunk (floating point: forward 1e-6, gradients 1e-5 relative to the gradient's scale)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(a, b, rel, what):
    scale = float(b.abs().max()) + 1e-30
    err = float((a - b).abs().max())
    assert err <= rel * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("shape", [(6, 64, 32, 32), (5, 16, 17, 30), (3, 512, 8, 8), (192, 8, 4, 4), (1, 3, 1, 1),
                                   (7, 5, 3, 5)])
@pytest.mark.parametrize("relu,with_res", [(True, False), (True, True), (False, False), (False, True)])
def test_bn_act_matches_stock_modules(cuda, shape, relu, with_res):
    from handobjectconsist_amd.nn import frozen_bn

    g = torch.Generator().manual_seed(hash((shape, relu, with_res)) % 2**31)
    N, C = shape[:2]
    bn = torch.nn.BatchNorm2d(C).to(cuda).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.randn(C, generator=g) * 0.5 + 1.0)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(C, generator=g) * 0.4)
        bn.running_var.copy_(torch.rand(C, generator=g) * 2 + 0.05)
    x = torch.randn(shape, generator=g).to(cuda).requires_grad_(True)
    res = torch.randn(shape, generator=g).to(cuda).requires_grad_(True) if with_res else None
    gy = torch.randn(shape, generator=g).to(cuda)
    y = frozen_bn.bn_act(x, bn, residual=res, relu=relu)
    y.backward(gy)
    got = [y.detach(), x.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone(), None if res is None else res.grad.clone()]
    x.grad = None
    bn.weight.grad = bn.bias.grad = None
    if res is not None:
        res.grad = None
    z = F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)
    if res is not None:
        z = z + res
    ref = F.relu(z) if relu else z
    ref.backward(gy)
    _close(got[0], ref.detach(), 2e-6, "forward")
    # a ReLU mask may differ where z rounds to the other side of 0 in the two formulations: compare off those pixels
    ambiguous = (z.detach().abs() < 1e-5) if relu else torch.zeros_like(z, dtype=torch.bool)
    _close(got[1].masked_fill(ambiguous, 0), x.grad.masked_fill(ambiguous, 0), 1e-5, "grad x")
    assert int(ambiguous.sum()) <= max(4, z.numel() // 20000)
    tol = 1e-4 if ambiguous.any() else 2e-5
    _close(got[2], bn.weight.grad, tol, "grad weight")
    _close(got[3], bn.bias.grad, tol, "grad bias")
    if res is not None:
        _close(got[4].masked_fill(ambiguous, 0), res.grad.masked_fill(ambiguous, 0), 1e-6, "grad residual")


def test_bn_act_rejects_training_mode_and_half(cuda):
    from handobjectconsist_amd.nn import frozen_bn

    bn = torch.nn.BatchNorm2d(4).to(cuda)
    x = torch.randn(2, 4, 8, 8, device=cuda)
    with pytest.raises(RuntimeError):
        frozen_bn.bn_act(x, bn)
    bn.eval()
    with pytest.raises(ValueError):
        frozen_bn.bn_act(x.half(), bn)  # fp16 is not an activation type of the kernels (fp32 and bf16 are)
    with pytest.raises(TypeError):
        frozen_bn.bn_act(x.cpu(), bn)
    assert frozen_bn.bn_act(x[:0], bn).shape == (0, 4, 8, 8)


def test_resnet_trunk_fused_equals_stock(cuda, monkeypatch):
    """The whole ResNet-18 trunk, forward features and every parameter gradient, fused BN path vs stock modules."""
    from handobjectconsist_amd.models import synthnet

    torch.manual_seed(0)
    net = synthnet.ResNet18Features().to(cuda).eval()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.1)
    x = torch.randn(6, 3, 96, 64, device=cuda)
    w = torch.randn(6, 512, device=cuda)
    out = {}
    for fused in (True, False):
        monkeypatch.setattr(synthnet, "USE_HIP_BN", fused)
        net.zero_grad(set_to_none=True)
        feats = net(x)
        (feats * w).sum().backward()
        out[fused] = (feats.detach().clone(), {n: p.grad.clone() for n, p in net.named_parameters()})
    _close(out[True][0], out[False][0], 1e-4, "features")
    for n, gref in out[False][1].items():
        _close(out[True][1][n], gref, 2e-3, f"grad {n}")


def test_resnet_trunk_fp16_autocast_takes_the_stock_modules(cuda, monkeypatch):
    """The fused glue is gated on the dtype the convolutions will PRODUCE: under fp16 autocast (neither fp32 nor
    bf16) the trunk must run on the stock modules instead of raising from the kernels' dtype check."""
    from handobjectconsist_amd.models import synthnet
    from handobjectconsist_amd import _lib

    calls = []
    real_call = _lib.call
    monkeypatch.setattr(_lib, "call", lambda name, *a: (calls.append(name), real_call(name, *a))[1])
    net = synthnet.ResNet18Features().to(cuda).eval()
    x = torch.randn(2, 3, 64, 64, device=cuda)
    with torch.autocast("cuda", dtype=torch.float16):
        feats = net(x)
    assert feats.shape == (2, 512) and torch.isfinite(feats.float()).all()
    assert not any(k.startswith(("mr_bn_act", "mr_stem_pool")) for k in calls), calls
    with torch.autocast("cuda", dtype=torch.bfloat16):
        net(x)
    assert "mr_stem_pool_forward" in calls and "mr_bn_act_forward" in calls, calls


@pytest.mark.parametrize("shape", [(4, 8, 32, 32), (3, 5, 17, 31), (2, 3, 1, 1), (2, 4, 135, 240), (6, 64, 64, 66),
                                   (1, 2, 2, 3)])
def test_stem_pool_matches_stock_modules(cuda, shape):
    """bn (frozen) -> relu -> MaxPool2d(3, 2, 1) as one kernel each way vs the three stock modules, incl. image
    sizes that are not multiples of the tile, 1-pixel images and a negative BatchNorm weight."""
    from handobjectconsist_amd.nn import frozen_bn

    g = torch.Generator().manual_seed(sum(shape))
    N, C = shape[:2]
    bn = torch.nn.BatchNorm2d(C).to(cuda).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.randn(C, generator=g) * 0.5 + 1.0)
        bn.weight[0] = -0.7
        bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(C, generator=g) * 0.4)
        bn.running_var.copy_(torch.rand(C, generator=g) * 2 + 0.05)
    x = torch.randn(shape, generator=g).to(cuda).requires_grad_(True)
    y = frozen_bn.stem_pool(x, bn)
    gy = torch.randn(y.shape, generator=g).to(cuda)
    y.backward(gy)
    got = (y.detach(), x.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone())
    x.grad = None
    bn.weight.grad = bn.bias.grad = None
    z = F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)
    ref = F.max_pool2d(F.relu(z), kernel_size=3, stride=2, padding=1)
    ref.backward(gy)
    assert got[0].shape == ref.shape
    _close(got[0], ref.detach(), 2e-6, "forward")
    # the arg-max of a window can differ where two candidates are within rounding of each other (different but
    # equivalent BN formulas): allow a handful of such windows, compare everything else tightly
    diff = (got[1] - x.grad).abs() > 1e-5 * float(x.grad.abs().max() + 1e-30)
    assert int(diff.sum()) <= max(2, x.numel() // 5000), int(diff.sum())
    tol = 2e-5 if not diff.any() else 5e-3
    _close(got[2], bn.weight.grad, tol, "grad weight")
    _close(got[3], bn.bias.grad, tol, "grad bias")
    # exact structural property: every pooled gradient lands on exactly one input pixel (or on none if relu is off there)
    assert float(got[1].abs().sum()) > 0 or float(gy.abs().sum()) == 0


@pytest.mark.parametrize("shape,with_res", [((6, 64, 32, 32), True), ((5, 16, 17, 30), False), ((3, 512, 8, 8), True),
                                            ((7, 5, 3, 5), True)])
def test_bn_act_bf16_activations(cuda, shape, with_res):
    """bf16 activations (the trunk under autocast): fp32 arithmetic on the converted values, one rounding on store --
    compared with the same computation in fp32 on the bf16-representable inputs."""
    from handobjectconsist_amd.nn import frozen_bn

    g = torch.Generator().manual_seed(sum(shape))
    C = shape[1]
    bn = torch.nn.BatchNorm2d(C).to(cuda).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.randn(C, generator=g) * 0.5 + 1.0)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(C, generator=g) * 0.4)
        bn.running_var.copy_(torch.rand(C, generator=g) * 2 + 0.05)
    x = torch.randn(shape, generator=g).to(cuda).bfloat16().requires_grad_(True)
    res = torch.randn(shape, generator=g).to(cuda).bfloat16().requires_grad_(True) if with_res else None
    gy = torch.randn(shape, generator=g).to(cuda).bfloat16()
    y = frozen_bn.bn_act(x, bn, residual=res, relu=True)
    assert y.dtype == torch.bfloat16
    y.backward(gy)
    xf = x.detach().float().requires_grad_(True)
    rf = res.detach().float().requires_grad_(True) if with_res else None
    gw, gb = bn.weight.grad.clone(), bn.bias.grad.clone()
    bn.weight.grad = bn.bias.grad = None
    ref = frozen_bn.bn_act(xf, bn, residual=rf, relu=True)   # the fp32 kernel (itself checked against stock PyTorch)
    ref.backward(gy.float())
    assert torch.equal(y, ref.detach().bfloat16())           # same arithmetic, one rounding to nearest-even
    assert torch.equal(x.grad, xf.grad.bfloat16()) and x.grad.dtype == torch.bfloat16
    if with_res:
        assert torch.equal(res.grad, rf.grad.bfloat16())
    _close(gw, bn.weight.grad, 1e-5, "grad weight")
    _close(gb, bn.bias.grad, 1e-5, "grad bias")


@pytest.mark.parametrize("shape", [(4, 8, 32, 32), (3, 5, 17, 31), (2, 4, 135, 240)])
def test_stem_pool_bf16_activations(cuda, shape):
    from handobjectconsist_amd.nn import frozen_bn

    g = torch.Generator().manual_seed(sum(shape))
    C = shape[1]
    bn = torch.nn.BatchNorm2d(C).to(cuda).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.randn(C, generator=g) * 0.5 + 1.0)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(C, generator=g) * 0.4)
        bn.running_var.copy_(torch.rand(C, generator=g) * 2 + 0.05)
    x = torch.randn(shape, generator=g).to(cuda).bfloat16().requires_grad_(True)
    y = frozen_bn.stem_pool(x, bn)
    gy = torch.randn(y.shape, generator=g).to(cuda).bfloat16()
    y.backward(gy)
    gw, gb = bn.weight.grad.clone(), bn.bias.grad.clone()
    bn.weight.grad = bn.bias.grad = None
    xf = x.detach().float().requires_grad_(True)
    ref = frozen_bn.stem_pool(xf, bn)
    ref.backward(gy.float())
    assert y.dtype == torch.bfloat16 and torch.equal(y, ref.detach().bfloat16())
    assert torch.equal(x.grad, xf.grad.bfloat16())
    _close(gw, bn.weight.grad, 1e-5, "grad weight")
    _close(gb, bn.bias.grad, 1e-5, "grad bias")


def test_resnet_trunk_bf16_autocast_fused_no_worse_than_stock(cuda, monkeypatch):
    """Whole trunk under bf16 autocast: rounding noise is amplified through 17 random-weight layers, so the fused and
    the stock bf16 runs are each compared with the fp32 run -- the fused path (which rounds once per group instead of
    after every element-wise module) must not be further from it than the stock bf16 path is."""
    from handobjectconsist_amd.models import synthnet

    torch.manual_seed(0)
    net = synthnet.ResNet18Features().to(cuda).eval()
    x = torch.randn(4, 3, 64, 64, device=cuda)

    def run(fused, autocast):
        monkeypatch.setattr(synthnet, "USE_HIP_BN", fused)
        net.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            feats = net(x)
        feats.float().sum().backward()
        return feats.detach().float(), net.conv1.weight.grad.clone()

    ref = run(False, False)
    stock, fused = run(False, True), run(True, True)
    for i, what in enumerate(("features", "stem weight gradient")):
        e_stock = float((stock[i] - ref[i]).abs().max())
        e_fused = float((fused[i] - ref[i]).abs().max())
        scale = float(ref[i].abs().max())
        assert e_fused <= 1.5 * e_stock + 1e-3 * scale, (what, e_fused, e_stock, scale)
        assert e_fused <= 0.2 * scale, (what, e_fused, scale)


def _bn(cuda, C, g):
    bn = torch.nn.BatchNorm2d(C).to(cuda).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.randn(C, generator=g) * 0.5 + 1.0)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(C, generator=g) * 0.4)
        bn.running_var.copy_(torch.rand(C, generator=g) * 2 + 0.05)
    return bn


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape,relu,with_res", [((6, 64, 32, 32), True, True), ((5, 16, 17, 30), True, False),
                                                 ((3, 512, 8, 8), True, True), ((7, 4, 3, 5), False, True),
                                                 ((2, 256, 1, 3), True, False), ((192, 8, 4, 4), False, False)])
def test_bn_act_channels_last_equals_nchw_kernel(cuda, shape, relu, with_res, dtype):
    """The channels-last kernels against the NCHW kernels (themselves checked against stock PyTorch): same arithmetic
    per element -> outputs and activation gradients bit-identical; the channel sums in a different order."""
    from handobjectconsist_amd.nn import frozen_bn

    g = torch.Generator().manual_seed(sum(shape))
    bn = _bn(cuda, shape[1], g)
    x0 = torch.randn(shape, generator=g).to(cuda).to(dtype)
    r0 = torch.randn(shape, generator=g).to(cuda).to(dtype) if with_res else None
    gy = torch.randn(shape, generator=g).to(cuda).to(dtype)
    outs = {}
    for cl in (False, True):
        fmt = torch.channels_last if cl else torch.contiguous_format
        x = x0.clone().contiguous(memory_format=fmt).requires_grad_(True)
        r = r0.clone().contiguous(memory_format=fmt).requires_grad_(True) if with_res else None
        bn.zero_grad(set_to_none=True)
        y = frozen_bn.bn_act(x, bn, residual=r, relu=relu)
        assert y.is_contiguous(memory_format=fmt) and y.dtype == dtype
        y.backward(gy.contiguous(memory_format=fmt))
        assert x.grad.is_contiguous(memory_format=fmt)
        outs[cl] = (y.detach(), x.grad, None if r is None else r.grad, bn.weight.grad.clone(), bn.bias.grad.clone())
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])
    if with_res:
        assert torch.equal(outs[True][2], outs[False][2])
    _close(outs[True][3], outs[False][3], 2e-5, "grad weight")
    _close(outs[True][4], outs[False][4], 2e-5, "grad bias")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(4, 8, 32, 32), (3, 64, 17, 31), (2, 4, 1, 1), (2, 16, 135, 240), (1, 128, 2, 3)])
def test_stem_pool_channels_last_equals_nchw_kernel(cuda, shape, dtype):
    from handobjectconsist_amd.nn import frozen_bn

    g = torch.Generator().manual_seed(sum(shape))
    bn = _bn(cuda, shape[1], g)
    with torch.no_grad():
        bn.weight[0] = -0.6
    x0 = torch.randn(shape, generator=g).to(cuda).to(dtype)
    outs = {}
    for cl in (False, True):
        fmt = torch.channels_last if cl else torch.contiguous_format
        x = x0.clone().contiguous(memory_format=fmt).requires_grad_(True)
        bn.zero_grad(set_to_none=True)
        y = frozen_bn.stem_pool(x, bn)
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(1)).to(cuda).to(dtype)
        y.backward(gy.contiguous(memory_format=fmt))
        outs[cl] = (y.detach(), x.grad, bn.weight.grad.clone(), bn.bias.grad.clone())
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])
    _close(outs[True][2], outs[False][2], 2e-5, "grad weight")
    _close(outs[True][3], outs[False][3], 2e-5, "grad bias")


def test_resnet_trunk_channels_last_equals_nchw(cuda, monkeypatch):
    """Whole trunk: channels-last (the default) vs NCHW activations, fused glue kernels in both -- MIOpen picks other
    convolution kernels per layout, so features / gradients agree to fp32 rounding, not bit for bit."""
    from handobjectconsist_amd.models import synthnet

    x = torch.randn(6, 3, 96, 64, device=cuda)
    w = torch.randn(6, 512, device=cuda)
    out = {}
    for cl in (True, False):
        monkeypatch.setattr(synthnet, "USE_CHANNELS_LAST", cl)
        torch.manual_seed(0)
        net = synthnet.ResNet18Features().to(cuda).eval()
        feats = net(x)
        (feats * w).sum().backward()
        assert net.conv1.weight.is_contiguous(memory_format=torch.channels_last) == cl or not cl
        out[cl] = (feats.detach().clone(), {n: p.grad.clone() for n, p in net.named_parameters()})
    _close(out[True][0], out[False][0], 1e-4, "features")
    for n, gref in out[False][1].items():
        _close(out[True][1][n], gref, 2e-3, f"grad {n}")


def test_trunk_glue_at_the_metric_shapes(cuda):
    """The shapes of a step at the metric configuration (3 x 64 frames of 256 x 256): stem [192,64,128,128] and a
    layer-1 group [192,64,64,64] -- channels-last against NCHW kernels bit for bit, plus size-independent properties
    (ReLU support, every pooled gradient lands on at most one input pixel, linearity of the backward in grad_y)."""
    from handobjectconsist_amd.nn import frozen_bn

    g = torch.Generator().manual_seed(0)
    bn = _bn(cuda, 64, g)
    x0 = torch.randn(192, 64, 128, 128, generator=g).to(cuda)
    outs = {}
    for cl in (False, True):
        fmt = torch.channels_last if cl else torch.contiguous_format
        x = x0.detach().clone(memory_format=fmt).requires_grad_(True)
        bn.zero_grad(set_to_none=True)
        y = frozen_bn.stem_pool(x, bn)
        gy = torch.ones_like(y)
        y.backward(gy)
        outs[cl] = (y.detach(), x.grad, bn.weight.grad.clone(), bn.bias.grad.clone())
    y, gx = outs[True][0], outs[True][1]
    assert torch.equal(y, outs[False][0]) and torch.equal(gx, outs[False][1])
    _close(outs[True][2], outs[False][2], 5e-5, "stem grad weight")
    assert y.shape == (192, 64, 64, 64) and float(y.min()) >= 0.0
    a = (bn.weight / torch.sqrt(bn.running_var + bn.eps))[None, :, None, None]
    # with grad_y = 1 every pooled value sends a * 1 to exactly one input pixel, unless the window is all <= 0
    routed = (gx / a).sum((2, 3))
    active = (y > 0).float().sum((2, 3))
    assert torch.allclose(routed, active, rtol=1e-5, atol=1e-2)
    del x0, outs, y, gx
    x1 = torch.randn(192, 64, 64, 64, generator=g).to(cuda)
    r1 = torch.randn(192, 64, 64, 64, generator=g).to(cuda)
    g1, g2 = torch.randn(192, 64, 64, 64, generator=g).to(cuda), torch.randn(192, 64, 64, 64, generator=g).to(cuda)
    res = {}
    for cl in (False, True):
        fmt = torch.channels_last if cl else torch.contiguous_format
        grads = []
        for gy in (g1, g2, g1 + g2):
            x = x1.detach().clone(memory_format=fmt).requires_grad_(True)
            r = r1.detach().clone(memory_format=fmt).requires_grad_(True)
            y = frozen_bn.bn_act(x, bn, residual=r, relu=True)
            y.backward(gy.contiguous(memory_format=fmt))
            grads.append((x.grad, r.grad))
        res[cl] = (y.detach(), grads)
    assert torch.equal(res[True][0], res[False][0])
    for k in range(3):
        assert torch.equal(res[True][1][k][0], res[False][1][k][0]) and torch.equal(res[True][1][k][1], res[False][1][k][1])
    # the backward is linear in grad_y; its identity branch passes grad_y through the ReLU mask unchanged
    _close(res[True][1][0][0] + res[True][1][1][0], res[True][1][2][0], 1e-6, "linearity of grad x")
    assert torch.equal(res[True][1][0][1], torch.where(res[True][0] > 0, g1, torch.zeros_like(g1)).contiguous(memory_format=torch.channels_last))


def test_training_trajectory_fused_trunk_vs_stock_modules(cuda, monkeypatch):
    """Ten optimiser steps of the whole trainer (data + consistency batches) with everything this build adds to the
    trunk (channels-last, fused BatchNorm / ReLU / identity / max-pool kernels, dual-output activations) against the
    stock NCHW modules: the loss trajectories stay together.  SGD and a zero weight on the photometric term: that
    term is piecewise constant in the network's outputs (pixel coverage, validity masks) and Adam normalises
    gradient magnitudes, so with either of them last-bit differences grow into different trajectories within a few
    steps -- in both builds alike; the smooth part of the objective is what can be compared step by step."""
    from handobjectconsist_amd.models import synthnet
    from handobjectconsist_amd.models.warpreg import WarpRegNet
    from handobjectconsist_amd.netscripts import epochpassconsist as E

    traj = {}
    for fused in (True, False):
        monkeypatch.setattr(synthnet, "USE_HIP_BN", fused)
        monkeypatch.setattr(synthnet, "USE_CHANNELS_LAST", fused)
        torch.manual_seed(0)
        model = synthnet.SynthMeshRegNet().to(cuda).eval()
        pre = WarpRegNet((64, 64), model, lambda_consist=0.0, lambda_data=1.0, criterion="l1", gt_refs=True,
                         use_backward=True, mano_faces=model.mano_layer.th_faces, pair_outputs="loss").to(cuda)
        pre.step_count = 1000
        opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-3)
        loader = E.SyntheticConsistLoader(4, 64, seed=0, device=cuda, pool=2)
        traj[fused] = [float(E.train_step(loader.step_batches(i), pre, opt)[0]) for i in range(10)]
        E.raise_pending_nan(opt)
    a, b = np.array(traj[True]), np.array(traj[False])
    assert np.all(np.isfinite(a)) and np.all(np.isfinite(b))
    assert np.abs(a - b).max() <= 2e-3 * np.abs(b).max(), (a, b)
    assert abs(b[-1] - b[0]) > 1e-6 * abs(b[0])  # the parameters do move


@pytest.mark.parametrize("channels_last", [False, True])
def test_glue_kernels_propagate_nan_like_the_stock_modules(cuda, channels_last):
    """torch.relu and max_pool2d propagate NaN; the fused kernels must too, or a diverged trunk would hand finite
    features to the 'Loss became nan!' guard of the epoch loop."""
    from handobjectconsist_amd.nn import frozen_bn

    bn = torch.nn.BatchNorm2d(8).to(cuda).eval()
    x = torch.randn(2, 8, 12, 12, device=cuda)
    x[0, 3, 5, 5] = float("nan")
    x[1, 0, 0, 0] = float("nan")
    if channels_last:
        x = x.contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        y = frozen_bn.bn_act(x, bn, relu=True)
        ref = F.relu(bn(x))
        assert torch.equal(torch.isnan(y), torch.isnan(ref)) and int(torch.isnan(y).sum()) == 2
        yp = frozen_bn.stem_pool(x, bn)
        refp = F.max_pool2d(F.relu(bn(x)), 3, 2, 1)
        assert torch.equal(torch.isnan(yp), torch.isnan(refp)) and int(torch.isnan(yp).sum()) >= 2
