"""A stand-in for MeshRegNet with three trainable scalars, shared by the fixture generator (which runs it under the
REFERENCE's WarpRegNet / epoch_pass on CPU, tests/golden/make_golden_trainer.py) and by the GPU test that runs it under
this package's counterparts: same ``(loss, results, losses)`` contract as meshregnet.py's forward -- a ``[1]`` loss,
results holding the predicted hand / object vertices of the frame, a losses dict with a ``None`` entry."""
import torch


class FakeMeshRegNet(torch.nn.Module):
    """forward(sample): sample[keys["pred_hand"]] + w0 * sample[keys["dir_hand"]] etc.  `keys` maps the names used
    here to the keys of the caller's sample dicts (the reference keys samples by its queries enums)."""

    def __init__(self, keys):
        super().__init__()
        self.keys = keys
        self.w = torch.nn.Parameter(torch.tensor([0.30, -0.20, 0.10]))

    def forward(self, sample):
        k = self.keys
        hand = sample[k["pred_hand"]] + self.w[0] * sample[k["dir_hand"]]
        obj = sample[k["pred_obj"]] + self.w[1] * sample[k["dir_obj"]]
        reg = ((self.w[2] * sample[k["reg_scale"]]) ** 2).sum().view(1)
        losses = {"mano_reg_loss": reg, "never_computed": None}
        loss = reg
        if k["gt_hand"] in sample and k["supervised"] in sample:
            l_hand = ((hand - sample[k["gt_hand"]]) ** 2).mean() * 1e4
            l_obj = ((obj - sample[k["gt_obj"]]) ** 2).mean() * 1e4
            losses["recov_joint3d"], losses["recov_objverts3d"] = l_hand, l_obj
            loss = loss + 0.5 * l_hand + 0.5 * l_obj
        losses["total_loss"] = loss
        return loss, {"recov_handverts3d": hand, "recov_objverts3d": obj}, losses
