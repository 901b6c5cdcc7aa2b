"""GraphedTrainStep (netscripts/epochpassconsist.py): one hipGraph launch per optimiser step must be the eager
``train_step`` -- same losses, same parameters after several steps (up to the order of the fp32 atomics of the render
backward), the lambda ramp followed, a NaN loss stopped on the device and raised by the next call."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _default_solvers():
    """GraphedTrainStep refuses MIOpen's solver search (torch.backends.cudnn.benchmark): see its docstring."""
    saved = torch.backends.cudnn.benchmark
    torch.backends.cudnn.benchmark = False
    yield
    torch.backends.cudnn.benchmark = saved


def _build(dev, B, is_, seed, capturable, lr=5e-5, pool=2):
    from handobjectconsist_amd.models.synthnet import SynthMeshRegNet
    from handobjectconsist_amd.models.warpreg import WarpRegNet
    from handobjectconsist_amd.netscripts.epochpassconsist import SyntheticConsistLoader

    torch.manual_seed(seed)
    model = SynthMeshRegNet().to(dev)
    model.eval()
    pre = WarpRegNet((is_, is_), model, lambda_consist=0.001, lambda_data=0.999, criterion="l1", gt_refs=True,
                     progressive_steps=6, use_backward=True, mano_faces=model.mano_layer.th_faces, pair_outputs="loss").to(dev)
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=lr, fused=True, capturable=capturable)
    loader = SyntheticConsistLoader(B, is_, seed=3, device=dev, pool=pool)
    return model, pre, opt, loader


def _grads(model):
    return torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None]).double()


def test_graph_replay_equals_the_eager_step(cuda):
    """Two identical models, learning rate 0 (the parameters stay put): every step is done eagerly on one (``train_step``) and
    through GraphedTrainStep on the other (two batch sets: one eager call each, then capture + replays) -- loss, every log
    entry and the gradients agree (the forward is deterministic up to the trunk's / heads' GEMM rounding, which the renderer
    amplifies for the consistency term; the gradients to the order of the render backward's fp32 atomics on top), the ramp is
    followed (progressive_steps = 6), the counters advance.  (One model per side: a premodel that has been through
    GraphedTrainStep keeps its AccumulateGrad nodes bound to that class's stream -- eager backward passes on the default
    stream in between would move them.)"""
    from handobjectconsist_amd.netscripts.epochpassconsist import GraphedTrainStep, raise_pending_nan, train_step

    B, is_, steps = 4, 128, 9
    _, pre_e, opt_e, loader_e = _build(cuda, B, is_, 11, True, lr=0.0)
    _, pre, opt, loader = _build(cuda, B, is_, 11, True, lr=0.0)
    step_g = GraphedTrainStep(pre, opt)
    losses_g = []
    for i in range(steps):
        assert pre_e.step_count == i and pre.step_count == i
        le, logs_e = train_step(loader_e.step_batches(i), pre_e, opt_e)
        grads_e = [p.grad.detach().clone() for g in opt_e.param_groups for p in g["params"]]
        lg, logs_g = step_g(loader.step_batches(i))
        assert pre.step_count == i + 1 and pre_e.step_count == i + 1
        losses_g.append(float(lg))
        assert set(logs_e) == set(logs_g)
        # (the consistency term of a RANDOM-INIT network: two eager calls differ by up to ~1e-3 -- the heads' GEMMs and the
        # trunk's convolutions are not bit-reproducible from call to call, the renderer's barycentrics amplify a last-bit change
        # of a few-pixel face by 10^3, and a pixel that changes sides of a validity threshold is 1e-3
        # of the masked mean of a small frame; its weight in the total is at most 0.001)
        consist_e = abs(float(logs_e["warp_consist"])) if "warp_consist" in logs_e else 0.0
        assert abs(float(lg) - float(le)) <= 1e-5 * abs(float(le)) + 0.001 * 1e-2 * consist_e + 1e-9, f"loss at step {i}"
        for k in logs_e:
            tol = 1e-2 if k == "warp_consist" else 2e-5
            np.testing.assert_allclose(float(logs_g[k]), float(logs_e[k]), rtol=tol, atol=1e-9, err_msg=f"{k} at step {i}")
        if i >= 2:  # a replayed step: its gradients live in the capture's own tensors
            assert step_g.replays == i - 1
            ge = torch.cat([g.flatten() for g in grads_e]).double()
            gg = torch.cat([g.flatten() for g in step_g.last_grads]).double()
            # (two eager calls on this workload: up to 3e-3 apart, the consistency term's share; a stale or missing
            # gradient shows as a difference of order one)
            assert float((ge - gg).norm() / ge.norm()) < 2e-2, f"gradients at step {i}"
    raise_pending_nan(opt)
    raise_pending_nan(opt_e)
    # the ramp is followed: the same batch set gives another loss while the weights still move (steps 0 / 2 / 4) ...
    assert abs(losses_g[0] - losses_g[2]) > 1e-7 and abs(losses_g[2] - losses_g[4]) > 1e-7
    assert abs(losses_g[6] - losses_g[8]) <= 1e-5 * abs(losses_g[8])  # ... and the same one once it is over
    assert all(float(st["step"]) == steps for st in opt.state_dict()["state"].values())


def test_replayed_update_equals_the_eager_update(cuda):
    """... and with trainmeshwarp.py's learning rate: from identical states one eager step, then one step -- replayed on
    one side, eager on the other -- moves the parameters by the same amounts (to the noise of the fp32 atomics in the
    gradients, which Adam's normalisation passes on: 2 % of the mean update)."""
    from handobjectconsist_amd.netscripts.epochpassconsist import GraphedTrainStep, train_step

    B, is_ = 4, 64
    m_e, pre_e, opt_e, ld_e = _build(cuda, B, is_, 21, False, pool=1)
    m_g, pre_g, opt_g, ld_g = _build(cuda, B, is_, 21, True, pool=1)
    step_g = GraphedTrainStep(pre_g, opt_g)
    train_step(ld_e.step_batches(0), pre_e, opt_e)
    step_g(ld_g.step_batches(0))
    before = [p.detach().clone() for p in m_g.parameters()]
    train_step(ld_e.step_batches(1), pre_e, opt_e)
    step_g(ld_g.step_batches(1))
    assert step_g.replays == 1
    torch.cuda.synchronize()
    num = den = 0.0
    for p_e, p_g, p0 in zip(m_e.parameters(), m_g.parameters(), before):
        if p_e.requires_grad:
            d_e, d_g = (p_e - p0).double(), (p_g - p0).double()
            num += float((d_e - d_g).detach().abs().sum()); den += float(d_e.detach().abs().sum())
    assert den > 0 and num / den < 0.02, num / den


def test_graph_replay_stops_a_nan_on_the_device(cuda):
    from handobjectconsist_amd.netscripts.epochpassconsist import GraphedTrainStep

    B, is_ = 2, 64
    model, pre, opt, loader = _build(cuda, B, is_, 5, True)
    step = GraphedTrainStep(pre, opt)
    for i in range(4):
        step(loader.step_batches(i))
    torch.cuda.synchronize()
    before = [p.detach().clone() for p in model.parameters()]
    image = loader.step_batches(4)[0]["data"][0]["image"]
    keep = image.clone()
    image.fill_(float("nan"))  # refilled IN PLACE: the graph reads the batch tensors where they are
    step(loader.step_batches(4))
    image.copy_(keep)
    with pytest.raises(ValueError, match="nan"):
        step(loader.step_batches(5))
    for a, b_ in zip(model.parameters(), before):
        assert torch.equal(a, b_), "the NaN step touched the parameters"
    step(loader.step_batches(5))  # ... and training goes on
    step(loader.step_batches(6))
    torch.cuda.synchronize()
    from handobjectconsist_amd.netscripts.epochpassconsist import raise_pending_nan
    raise_pending_nan(opt)
    assert any(not torch.equal(a, b_) for a, b_ in zip(model.parameters(), before))
