"""The numpy warp oracle against the golden vectors captured from the REAL reference
(tests/golden/make_golden_warp.py imports /root/reference/meshreg/warping/imgflowarp.py etc.)."""
import os

import numpy as np

from oracle import warp_ref as W

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_warp_bilinear_nearest_and_grads():
    g = np.load(os.path.join(GOLDEN, "warp_basic.npz"))
    for mode in ("bilinear", "nearest"):
        out, mask = W.warp(g["x"], g["flow"], mode=mode)
        assert (mask != g[f"mask_{mode}"]).sum() == 0
        assert np.abs(out - g[f"out_{mode}"]).max() < 2e-6
    gflow, gx = W.warp_backward(g["x"], g["flow"], g["grad_out"])
    assert np.abs(gflow - g["grad_flow"]).max() < 1e-5 * max(1, np.abs(g["grad_flow"]).max())
    assert np.abs(gx - g["grad_x"]).max() < 1e-5
    # flows include exactly-zero, integer, sub-pixel and far out-of-bounds vectors
    assert (g["flow"] == 0).any() and (np.abs(g["flow"]) > 500).any() and g["mask_bilinear"].min() == 0


def test_meshgrid():
    g = np.load(os.path.join(GOLDEN, "warp_basic.npz"))
    for s in (0, 1):
        assert np.array_equal(W.get_spatial_meshgrid(g["x"].shape, bool(s)), g[f"meshgrid_{s}"])


def test_occlusion_mask():
    g = np.load(os.path.join(GOLDEN, "warp_occlusion.npz"))
    o1, o2 = W.get_occlusion_mask(g["mask_flow1"], g["mask_flow2"], g["flow12"], g["flow21"])
    assert np.abs(o1 - g["occl1"]).max() < 1e-7 and np.abs(o2 - g["occl2"]).max() < 1e-7
    assert g["occl1"].sum() > 10 and (g["occl1"] == 0).any()


def test_pair_consist_loss_masks_grads():
    g = np.load(os.path.join(GOLDEN, "warp_pair_consist.npz"))
    flows = [g["flow12"], g["flow21"]]
    args = (g["image_ref"], g["image"], g["jitter_ref"], g["jitter"])
    for ub in (0, 1):
        loss, masks, warps, diffs, _ = W.pair_consist(flows, *args, bool(ub))
        assert np.abs(loss - g[f"loss_ub{ub}"]).max() < 1e-6
        grads = W.pair_consist_grad(flows, *args, g["grad_loss"], bool(ub))
        for k, name in ((0, "grad_flow12"), (1, "grad_flow21")):
            ref = g[f"{name}_ub{ub}"]
            assert np.abs(grads[k] - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-12) + 1e-9
    for i in (1, 2):
        assert (masks[i - 1]["warp_mask"] != g[f"warp_mask{i}"]).sum() == 0
        assert (masks[i - 1]["full_mask"] != g[f"full_mask{i}"]).sum() == 0
        assert (masks[i - 1]["flow_mask"] != g[f"flow_mask{i}"]).sum() == 0
        assert np.abs(warps[i - 1] - g[f"warp{i}"]).max() < 2e-6
        assert np.abs(diffs[i - 1] - g[f"diff{i}"]).max() < 2e-6
    assert g["loss_ub1"][2] == 0  # a sample with no valid pixel: masked mean divides by 1


def test_masked_mean():
    g = np.load(os.path.join(GOLDEN, "warp_misc.npz"))
    assert np.abs(W.batch_masked_mean_loss(g["dists"], g["mask"]) - g["masked_mean"]).max() < 1e-6
    assert g["masked_mean"][1] == 0
