"""Error report of the HIP chain against the reference-glue fixtures (tests/golden/chain_*.npz): the numbers
behind tests/test_gpu_chain.py's tolerances.  GPU box:  python scripts/chain_report.py > gpurun_out/chain_report.json"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_chain as T  # noqa: E402

from handobjectconsist_amd.models import warpbranch  # noqa: E402
from handobjectconsist_amd.optim.pyramidloss import PyramidCriterion  # noqa: E402
from handobjectconsist_amd.warping import opticalflow  # noqa: E402

dev = torch.device("cuda:0")
rep = {"opticalflow": [], "warpbranch": []}
z, meta = T.load("chain_opticalflow.npz")
for m in meta:
    s, k, is_ = m["scene"], m["key"], m["image_size"]
    v1, v2 = T.t(z[f"{s}_verts1"], dev, True), T.t(z[f"{s}_verts2"], dev, True)
    flows = opticalflow.get_opticalflow(
        [v1, v2], T.t(z[f"{s}_faces"], dev), [T.t(z[f"{s}_K1"], dev), T.t(z[f"{s}_K2"], dev)],
        T._training_renderer(is_, dev), orig_img_size=m["orig_img_size"], mask_occlusions=m["mask_occlusions"],
        detach_textures=m["detach_textures"], detach_renders=m["detach_renders"],
        ignore_face_idxs=m["ignore_face_idxs"] if m["ignore"] else None)
    ((flows[0] * T.t(z[f"{s}_g12"], dev)).sum() + (flows[1] * T.t(z[f"{s}_g21"], dev)).sum()).backward()
    row = {"key": k}
    for i, name in enumerate(("flow12", "flow21")):
        got, want = T.n(flows[i]), z[f"{k}_{name}"]
        row[f"{name}_support_mismatch"] = int(((got != 0) != (want != 0)).sum())
        row[f"{name}_max_abs_err"] = float(np.abs(got - want).max())
    for v, name in ((v1, "grad_verts1"), (v2, "grad_verts2")):
        want = z[f"{k}_{name}"]
        got = T.n(v.grad) if v.grad is not None else np.zeros_like(want)
        row[f"{name}_norm_rel"] = float(T.norm_rel(got, want)) if np.abs(want).max() > 0 else float(np.abs(got).max())
    rep["opticalflow"].append(row)
z, meta = T.load("chain_warpbranch.npz")
for m in meta:
    k, is_, crop = m["key"], m["image_size"], tuple(m["input_res"])
    samples, results = [], []
    for f in range(m["frames"]):
        samples.append({"image": T.t(z[f"f{f}_image"], dev), "jittermask": T.t(z[f"f{f}_jittermask"], dev),
                        "camintr": T.t(z[f"f{f}_camintr"], dev), "objfaces": T.t(z[f"f{f}_objfaces"], dev),
                        "objverts3d": T.t(z[f"f{f}_gt_obj"], dev), "handverts3d": T.t(z[f"f{f}_gt_hand"], dev)})
        results.append({"recov_handverts3d": T.t(z[f"f{f}_pred_hand"], dev, True),
                        "recov_objverts3d": T.t(z[f"f{f}_pred_obj"], dev, True)})
    loss, pair = warpbranch.forward(samples, results, T.t(z["hand_face"], dev)[None], T._training_renderer(is_, dev), crop,
                                    PyramidCriterion("l1"), gt_refs=m["gt_refs"], first_only=m["first_only"],
                                    hand_ignore_faces=m["hand_ignore_faces"], use_backward=m["use_backward"])
    loss.backward()
    row = {"key": k, "loss_rel": abs(float(loss.detach()) - float(z[f"{k}_loss"])) / abs(float(z[f"{k}_loss"]))}
    row["support_mismatch"] = sum(int(((T.n(pair["recons_flows"][p][d]) != 0) != (z[f"{k}_p{p}_flow{d}"] != 0)).sum())
                                  for p in range(m["frames"] - 1) for d in (0, 1))
    for f, res in enumerate(results):
        for name in ("recov_handverts3d", "recov_objverts3d"):
            want = z[f"{k}_f{f}_grad_{name}"]
            got = T.n(res[name].grad) if res[name].grad is not None else np.zeros_like(want)
            row[f"f{f}_{name}_norm_rel"] = float(T.norm_rel(got, want)) if np.abs(want).max() > 0 else float(np.abs(got).max())
    rep["warpbranch"].append(row)
print(json.dumps(rep, indent=1))
