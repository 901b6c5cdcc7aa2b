"""GraphedTrainStep (netscripts/epochpassconsist.py): one hipGraph launch per optimiser step must be the eager
``train_step`` -- same losses, same parameters after several steps (up to the order of the fp32 atomics of the render
backward), the lambda ramp followed, a NaN loss stopped on the device and raised by the next call."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(dev, B, is_, seed, capturable):
    from handobjectconsist_amd.models.synthnet import SynthMeshRegNet
    from handobjectconsist_amd.models.warpreg import WarpRegNet
    from handobjectconsist_amd.netscripts.epochpassconsist import SyntheticConsistLoader

    torch.manual_seed(seed)
    model = SynthMeshRegNet().to(dev)
    model.eval()
    pre = WarpRegNet((is_, is_), model, lambda_consist=0.001, lambda_data=0.999, criterion="l1", gt_refs=True,
                     progressive_steps=6, use_backward=True, mano_faces=model.mano_layer.th_faces, pair_outputs="loss").to(dev)
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=5e-5, fused=True, capturable=capturable)
    loader = SyntheticConsistLoader(B, is_, seed=3, device=dev, pool=2)
    return model, pre, opt, loader


def test_graph_replay_equals_the_eager_step(cuda):
    from handobjectconsist_amd.netscripts.epochpassconsist import GraphedTrainStep, raise_pending_nan, train_step

    B, is_, steps = 4, 64, 9
    m_e, pre_e, opt_e, ld_e = _build(cuda, B, is_, 11, False)
    m_g, pre_g, opt_g, ld_g = _build(cuda, B, is_, 11, True)
    for a, b_ in zip(m_e.parameters(), m_g.parameters()):
        assert torch.equal(a, b_)
    step_g = GraphedTrainStep(pre_g, opt_g)
    losses_e, losses_g = [], []
    for i in range(steps):
        le, logs_e = train_step(ld_e.step_batches(i), pre_e, opt_e)
        lg, logs_g = step_g(ld_g.step_batches(i))
        losses_e.append(float(le)); losses_g.append(float(lg))
        assert set(logs_e) == set(logs_g)
    raise_pending_nan(opt_e); raise_pending_nan(opt_g)
    assert step_g.replays == steps - 2, "two batch sets: one eager call each, then replays"
    assert pre_e.step_count == pre_g.step_count == steps
    # the ramp (progressive_steps = 6) moves the weights during the first steps: frozen weights would show here
    np.testing.assert_allclose(losses_g, losses_e, rtol=2e-5, atol=1e-7)
    for (name, a), b_ in zip(m_e.named_parameters(), m_g.parameters()):
        if a.requires_grad:
            scale = float(a.abs().max()) + 1e-8
            assert float((a - b_).abs().max()) <= 2e-5 * scale + 2e-7, name
    sd_e, sd_g = opt_e.state_dict()["state"], opt_g.state_dict()["state"]
    assert all(float(sd_e[k]["step"]) == float(sd_g[k]["step"]) == steps for k in sd_e)


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
