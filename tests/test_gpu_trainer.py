"""The trainer glue and the METRIC workload against fixtures produced by running the reference's own code
(tests/golden/make_golden_trainer.py, build container): ``WarpRegNet.forward`` (warpreg.py:81-127) and the two-batch
accumulation of ``epoch_pass`` (epochpassconsist.py:56-68) around a three-parameter stand-in for MeshRegNet
(tests/trainer_fake.py), and ``get_opticalflow`` + ``warpbranch.forward`` on the bench's own meshes (778-vertex hand +
1002-vertex object, 7104 faces after fill-back) at 256 x 256."""
import json
import os

import numpy as np
import pytest
import torch

from trainer_fake import FakeMeshRegNet

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name))
    return z, json.loads(str(z["meta"]))


def t(a, dev, grad=False):
    x = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return x.requires_grad_(True) if grad else x


def n(x):
    return x.detach().cpu().numpy()


def norm_rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _renderer(is_, dev):
    from handobjectconsist_amd.neurender.renderer import Renderer

    return Renderer(image_size=is_, R=torch.eye(3, device=dev).unsqueeze(0), t=torch.zeros(1, 3, device=dev),
                    K=torch.ones(1, 3, 3, device=dev), orig_size=is_, anti_aliasing=False, fill_back=True, near=0.1,
                    no_light=True)


def _check_flow(z, key, flow):
    """Seeded sample (exact support, 1e-6), support count and sum of a flow the fixture stores in summary."""
    got = n(flow).reshape(-1, 2)
    idx, want = z[f"{key}_idx"], z[f"{key}_sample"]
    assert int((got[:, 0] != 0).sum()) == int(z[f"{key}_support"][0]), (key, "support")
    assert np.array_equal(got[idx] != 0, want != 0), (key, "sampled support")
    assert np.abs(got[idx] - want).max() <= 1e-6 * max(np.abs(want).max(), 1.0), (key, np.abs(got[idx] - want).max())
    assert np.allclose(got.astype(np.float64).sum(0), z[f"{key}_sum"], rtol=1e-5, atol=1e-3), (key, "sum")


# ---------------------------------------------------------------------------------------------------
# the metric workload
# ---------------------------------------------------------------------------------------------------


def _metric_scene(z, m):
    """The fixture's inputs ARE the bench's generators' output for the stored seed (checked by their sums)."""
    from handobjectconsist_amd.utils import synth

    s = synth.random_scene(m["batch"], seed=m["scene_seed"], image_size=m["image_size"])
    im_ref, im, jm_ref, jm = synth.random_images(m["batch"], m["image_size"], m["image_size"], m["scene_seed"])
    got = dict(verts1=s["verts1"], verts2=s["verts2"], K1=s["K1"], faces=s["faces"], image0=im, image1=im_ref, jitter0=jm, jitter1=jm_ref)
    for k, v in got.items():
        assert float(np.asarray(v, np.float64).sum()) == float(z["checksum_" + k]), f"input {k} differs from the fixture's"
    return s, (im, im_ref), (jm, jm_ref)



@pytest.mark.parametrize("poisoned", [False, True])
def test_metric_workload_get_opticalflow_against_reference_glue(cuda, poisoned):
    """The bench's meshes (utils/synth.random_scene) at 256 x 256 through the stacked training node, against the
    reference's get_opticalflow: flows (exact support at 40 000 seeded pixels, values 1e-6, support counts, sums) and
    d / d vertices of BOTH frames in full at the north-star tolerance."""
    from handobjectconsist_amd.utils import synth
    from handobjectconsist_amd.warping import opticalflow

    z, m = load("chain_metric.npz")
    B, is_ = m["batch"], m["image_size"]
    s = _metric_scene(z, m)[0]
    assert s["faces"].shape[1] * 2 == 7104
    saved = opticalflow.DEBUG_POISON_RENDER_OUTPUTS
    opticalflow.DEBUG_POISON_RENDER_OUTPUTS = poisoned
    try:
        v1, v2 = t(s["verts1"], cuda, True), t(s["verts2"], cuda, True)
        flows = opticalflow.get_opticalflow([v1, v2], t(s["faces"], cuda), [t(s["K1"], cuda), t(s["K2"], cuda)],
                                            _renderer(is_, cuda), orig_img_size=(is_, is_), mask_occlusions=True,
                                            detach_textures=False, detach_renders=True,
                                            ignore_face_idxs=m["hand_ignore_faces"])
        assert hasattr(flows[0]._base, "_hoc_coverage"), "the stacked training node must have been taken"
        _check_flow(z, "of_flow12", flows[0])
        _check_flow(z, "of_flow21", flows[1])
        r = np.random.default_rng(m["grad_seed"])
        g12 = r.standard_normal((B, is_, is_, 2)).astype(np.float32)
        g21 = r.standard_normal((B, is_, is_, 2)).astype(np.float32)
        ((flows[0] * t(g12, cuda)).sum() + (flows[1] * t(g21, cuda)).sum()).backward()
        for v, name in ((v1, "of_grad_verts1"), (v2, "of_grad_verts2")):
            assert np.abs(z[name]).max() > 0
            assert norm_rel(n(v.grad), z[name]) < 1e-4, (name, norm_rel(n(v.grad), z[name]))
    finally:
        opticalflow.DEBUG_POISON_RENDER_OUTPUTS = saved


def test_metric_workload_warpbranch_against_reference_glue(cuda):
    """... and through warpbranch.forward in the trainer's setting (gt_refs, first_only, use_backward): per-sample pair
    losses, the loss, valid-pixel counts, flows, d loss / d predicted vertices of frame 0; frame 1 receives nothing."""
    from handobjectconsist_amd.models import warpbranch
    from handobjectconsist_amd.optim.pyramidloss import PyramidCriterion

    z, m = load("chain_metric.npz")
    B, is_ = m["batch"], m["image_size"]
    s, images, jitters = _metric_scene(z, m)
    samples, results = [], []
    for k in (0, 1):
        f = "12"[k]
        samples.append({"image": t(images[k], cuda), "jittermask": t(jitters[k], cuda), "camintr": t(s["K" + f], cuda),
                        "objfaces": t(s["obj_faces"][None].repeat(B, 0), cuda), "objverts3d": t(s["obj_verts" + f], cuda),
                        "handverts3d": t(s["hand_verts" + f], cuda)})
    results.append({"recov_handverts3d": t(s["hand_verts1"], cuda, True), "recov_objverts3d": t(s["obj_verts1"], cuda, True)})
    results.append({"recov_handverts3d": t(z["pred1_hand"], cuda, True), "recov_objverts3d": t(z["pred1_obj"], cuda, True)})
    loss, pair = warpbranch.forward(samples, results, t(s["hand_faces"], cuda)[None], _renderer(is_, cuda), (is_, is_),
                                    PyramidCriterion("l1"), gt_refs=True, first_only=True,
                                    hand_ignore_faces=m["hand_ignore_faces"], use_backward=True)
    loss.backward()
    for d in (0, 1):
        _check_flow(z, f"wb_flow{d}", pair["recons_flows"][0][d])
        assert float(pair["masks"][0][d]["full_mask"].sum()) == float(z[f"wb_full_mask{d}_sum"]), ("valid pixels", d)
    assert norm_rel(n(pair["diff_losses"]), z["wb_diff_losses"]) < 1e-5
    assert abs(float(loss) - float(z["wb_loss"])) < 1e-5 * abs(float(z["wb_loss"]))
    for name, key in (("recov_handverts3d", "wb_grad_hand0"), ("recov_objverts3d", "wb_grad_obj0")):
        assert np.abs(z[key]).max() > 0
        assert norm_rel(n(results[0][name].grad), z[key]) < 1e-4, (key, norm_rel(n(results[0][name].grad), z[key]))
        assert results[1][name].grad is None, "the annotated frame must not receive a gradient"


@pytest.mark.parametrize("mode", ["step", "unit", "recompute"])
def test_metric_workload_training_mode_against_the_reference(cuda, monkeypatch, mode):
    """The same reference run against warpbranch.forward AS THE TRAINER CALLS IT (pair_outputs="loss"): the pair is one
    fused node (opticalflow.flow_pair_loss) -- through the two struct calls of ABI 8 (``step``: what training runs), as the
    node pair of rounds 4-5 with the pair loss's gradient formed by the forward launch (``unit``) or recomputed by the
    backward launch (``recompute``) -- on NaN-poisoned buffers.  Loss, per-sample pair losses and
    d loss / d predicted vertices of frame 0 are the reference's; the flows are, wherever the reference's are non-zero."""
    from handobjectconsist_amd import _lib
    from handobjectconsist_amd.models import warpbranch
    from handobjectconsist_amd.optim.pyramidloss import PyramidCriterion
    from handobjectconsist_amd.warping import opticalflow, pairstep

    z, m = load("chain_metric.npz")
    B, is_ = m["batch"], m["image_size"]
    s, images, jitters = _metric_scene(z, m)
    unit = mode != "recompute"
    monkeypatch.setattr(opticalflow, "USE_UNIT_GRADIENT", unit)
    monkeypatch.setattr(opticalflow, "USE_PAIR_STEP", mode == "step")
    monkeypatch.setattr(opticalflow, "DEBUG_POISON_RENDER_OUTPUTS", True)
    calls = []
    real_call, real_step = _lib.call, pairstep.pair_step
    monkeypatch.setattr(_lib, "call", lambda name, *a: (calls.append(name), real_call(name, *a))[1])
    monkeypatch.setattr(pairstep, "pair_step", lambda *a, **k: (calls.append("pair_step"), real_step(*a, **k))[1])
    samples, results = [], []
    for k in (0, 1):
        f = "12"[k]
        samples.append({"image": t(images[k], cuda), "jittermask": t(jitters[k], cuda), "camintr": t(s["K" + f], cuda),
                        "objfaces": t(s["obj_faces"][None].repeat(B, 0), cuda), "objverts3d": t(s["obj_verts" + f], cuda),
                        "handverts3d": t(s["hand_verts" + f], cuda)})
    results.append({"recov_handverts3d": t(s["hand_verts1"], cuda, True), "recov_objverts3d": t(s["obj_verts1"], cuda, True)})
    results.append({"recov_handverts3d": t(z["pred1_hand"], cuda, True), "recov_objverts3d": t(z["pred1_obj"], cuda, True)})
    loss, pair = warpbranch.forward(samples, results, t(s["hand_faces"], cuda)[None], _renderer(is_, cuda), (is_, is_),
                                    PyramidCriterion("l1"), gt_refs=True, first_only=True,
                                    hand_ignore_faces=m["hand_ignore_faces"], use_backward=True, pair_outputs="loss")
    loss.backward()
    fwd, bwd = (("mr_flow_pair_forward_grad_tiles", "mr_flow_pair_backward_unit_tiles") if unit
                else ("mr_flow_pair_forward_tiles", "mr_flow_pair_backward_tiles"))
    if mode == "step":
        assert "pair_step" in calls and fwd not in calls, "the trainer's setting must take the struct path"
    else:
        assert fwd in calls and bwd in calls and "pair_step" not in calls, "the trainer's setting must take the fused pair node"
    assert all(x is None for x in pair["masks"]) and all(x is None for x in pair["warps"])  # nothing per-pixel in this mode
    for d in (0, 1):
        got = n(pair["recons_flows"][0][d]).reshape(-1, 2)
        idx, want = z[f"wb_flow{d}_idx"], z[f"wb_flow{d}_sample"]
        on = want[:, 0] != 0  # (defined under the covered tiles only; the reference's flow is zero elsewhere)
        assert on.any() and np.abs(got[idx][on] - want[on]).max() <= 1e-6 * max(np.abs(want).max(), 1.0), ("flow", d)
    assert norm_rel(n(pair["diff_losses"]), z["wb_diff_losses"]) < 1e-5
    assert abs(float(loss) - float(z["wb_loss"])) < 1e-5 * abs(float(z["wb_loss"]))
    for name, key in (("recov_handverts3d", "wb_grad_hand0"), ("recov_objverts3d", "wb_grad_obj0")):
        assert norm_rel(n(results[0][name].grad), z[key]) < 1e-4, (key, norm_rel(n(results[0][name].grad), z[key]))
        assert results[1][name].grad is None, "the annotated frame must not receive a gradient"


# ---------------------------------------------------------------------------------------------------
# WarpRegNet.forward and epoch_pass
# ---------------------------------------------------------------------------------------------------

_KEYS = dict(pred_hand="_pred_hand", dir_hand="_dir_hand", pred_obj="_pred_obj", dir_obj="_dir_obj", reg_scale="_reg_scale",
             gt_hand="handverts3d", gt_obj="objverts3d", supervised="_supervised")


def _trainer_batches(z, cfg, dev):
    """tests/golden/make_golden_trainer.py::trainer_batches with this package's string keys: scenes / images from the
    seeded generators (sums checked against the fixture), the stand-in model's per-frame inputs from the fixture."""
    from handobjectconsist_amd.utils import synth

    B, is_ = cfg["batch"], cfg["image_size"]
    batches = []
    for step in range(cfg["steps"]):
        s = synth.random_scene(B, seed=300 + step, image_size=is_)
        im_ref, im, jm_ref, jm = synth.random_images(B, is_, is_, 300 + step)
        assert float(np.float64(s["verts1"].sum()) + np.float64(im.sum()) + np.float64(jm_ref.sum())) == float(z[f"s{step}_checksum"])

        def frame(tag, hand, obj, K, img, jit, supervised):
            g = lambda k: t(z[f"s{step}_{tag}_{k}"], dev)
            sample = {"image": t(img, dev), "jittermask": t(jit, dev), "camintr": t(K, dev),
                      "objfaces": t(s["obj_faces"][None].repeat(B, 0), dev), "_pred_hand": g("pred_hand"),
                      "_dir_hand": g("dir_hand"), "_pred_obj": g("pred_obj"), "_dir_obj": g("dir_obj"), "_reg_scale": g("reg_scale")}
            if supervised:
                sample.update({"handverts3d": t(hand, dev), "objverts3d": t(obj, dev), "_supervised": True})
            return sample

        batches.append({"data": [frame("d", s["hand_verts2"], s["obj_verts2"], s["K2"], im_ref, jm_ref, True)], "supervision": "data"})
        batches.append({"data": [frame("c0", s["hand_verts1"], s["obj_verts1"], s["K1"], im, jm, False),
                                 frame("c1", s["hand_verts2"], s["obj_verts2"], s["K2"], im_ref, jm_ref, True)],
                        "supervision": "consist"})
    return batches


def _build(cfg, dev):
    from handobjectconsist_amd.models.warpreg import WarpRegNet
    from handobjectconsist_amd.utils import synth

    model = FakeMeshRegNet(_KEYS).to(dev)
    is_ = cfg["image_size"]
    pre = WarpRegNet((is_, is_), model, lambda_data=cfg["lambda_data"], lambda_consist=cfg["lambda_consist"], criterion="l1",
                     progressive_steps=cfg["progressive_steps"], use_backward=True, gt_refs=True,
                     mano_faces=torch.from_numpy(synth.hand_template()[1][:1538].copy())).to(dev)
    assert pre.hand_ignore_faces == cfg["hand_ignore_faces"]
    return model, pre


def test_warpregnet_forward_against_the_reference(cuda):
    """warpreg.py:81-127 run from the reference: total loss, every aggregate loss (names taken from the FIRST sample,
    ``None`` entries skipped, ``reg_loss`` / ``warp_consist`` added), d loss / d parameters and ``step_count`` at five
    points of the lambda ramp, for a data batch and a consist batch."""
    z, cfg = load("chain_trainer.npz")
    model, pre = _build(cfg, cuda)
    batches = _trainer_batches(z, cfg, cuda)
    for step_count in (0, 1, 2, 3, 7):
        for batch in batches[:2]:
            key = f"fw_{batch['supervision']}_{step_count}"
            pre.step_count = step_count
            model.zero_grad()
            loss, agg, results, pair = pre.forward(batch)
            loss.sum().backward()
            assert (pair is None) == (batch["supervision"] == "data")
            assert len(results) == len(batch["data"])
            assert pre.step_count == int(z[f"{key}_step_count_after"]), key
            assert sorted(agg) == json.loads(str(z[f"{key}_agg_names"])), (key, sorted(agg))
            assert norm_rel(n(loss).reshape(-1), z[f"{key}_loss"]) < 1e-5, (key, float(loss.sum()), z[f"{key}_loss"])
            for name, val in agg.items():
                assert norm_rel(n(val).reshape(-1), z[f"{key}_agg_{name}"]) < 1e-5, (key, name)
            assert norm_rel(n(model.w.grad), z[f"{key}_grad_w"]) < 1e-4, (key, n(model.w.grad), z[f"{key}_grad_w"])


@pytest.mark.parametrize("fused_optimizer", [False, True])
def test_epoch_pass_accumulation_against_the_reference(cuda, fused_optimizer):
    """epochpassconsist.py:56-68 run from the reference (4 optimiser steps of SGD over alternating data / consist
    batches, loader_nb = 2, the lambda ramp running): the parameters after EVERY step and the final step_count."""
    from handobjectconsist_amd.netscripts.epochpassconsist import epoch_pass, train_step

    z, cfg = load("chain_trainer.npz")
    model, pre = _build(cfg, cuda)
    batches = _trainer_batches(z, cfg, cuda)
    assert np.allclose(n(model.w), z["ep_w_before"])
    opt = torch.optim.SGD(model.parameters(), lr=cfg["lr"], **({"fused": True} if fused_optimizer else {}))
    want = z["ep_w_after_each_step"]
    if fused_optimizer:  # the whole epoch through epoch_pass (device-side NaN guard path)
        history = epoch_pass(batches, pre, opt, loader_nb=2)
        assert len(history) == cfg["steps"] and all(torch.isfinite(h) for h in history)
        assert norm_rel(n(model.w), want[-1]) < 1e-4, (n(model.w), want[-1])
    else:
        for step in range(cfg["steps"]):
            train_step(batches[2 * step:2 * step + 2], pre, opt)
            assert norm_rel(n(model.w), want[step]) < 1e-4, (step, n(model.w), want[step])
    assert pre.step_count == int(z["ep_step_count_after"])


def test_eager_step_follows_the_ramp_with_lambda_tensors(cuda):
    """A premodel that carries the device-side lambda tensors (`refresh_lambda_tensors`: what a captured step reads its two
    weights from -- scripts/graph_step_experiment.py) and is stepped eagerly must read the weights of ITS step: the eager
    forward refreshes the tensor (warpreg.py:103-110's ramp, one value per step)."""
    from handobjectconsist_amd.models.synthnet import SynthMeshRegNet
    from handobjectconsist_amd.models.warpreg import WarpRegNet
    from handobjectconsist_amd.netscripts.epochpassconsist import SyntheticConsistLoader, train_step

    saved = torch.backends.cudnn.benchmark
    torch.backends.cudnn.benchmark = False  # (MIOpen's default solver choice for both twins)
    try:
        twins = []
        for _ in range(2):
            torch.manual_seed(7)
            model = SynthMeshRegNet().to(cuda)
            model.eval()
            pre = WarpRegNet((64, 64), model, lambda_consist=0.001, lambda_data=0.999, criterion="l1", gt_refs=True,
                             progressive_steps=6, use_backward=True, mano_faces=model.mano_layer.th_faces, pair_outputs="loss").to(cuda)
            opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=0.0, fused=True)
            twins.append((pre, opt, SyntheticConsistLoader(2, 64, seed=3, device=cuda, pool=2)))
        (pre_a, opt_a, ld_a), (pre_b, opt_b, ld_b) = twins
        pre_b.refresh_lambda_tensors()
        for i in range(4):
            la, _ = train_step(ld_a.step_batches(i), pre_a, opt_a)
            lb, _ = train_step(ld_b.step_batches(i), pre_b, opt_b)
            consist = 0.3  # (upper bound of the term at random init; its weight is at most 0.001 and it scatters by ~1e-3)
            assert abs(float(la) - float(lb)) <= 1e-5 * abs(float(la)) + 0.001 * 1e-2 * consist, f"step {i}"
    finally:
        torch.backends.cudnn.benchmark = saved
