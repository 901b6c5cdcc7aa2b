"""Photometric criterion -- drop-in for meshreg/optim/pyramidloss.py.

Only the configuration the reference ever enables is on the hot path:
``PyramidCriterion('l1')`` with ``level_nb=1`` (warpreg.py:34, trainmeshwarp.py
``--consist_criterion l1``); ``imgflowarp.pair_consist`` recognises it and runs the fused HIP
kernel.  ``l2`` works through the composed path.  ``ssim`` and ``level_nb > 1`` need kornia's
SSIM / ScalePyramid (third-party, never enabled by any reference script): out of scope, they
raise NotImplementedError instead of silently computing something else."""
import torch

from handobjectconsist_amd.optim import lossutils


class PyramidCriterion:
    def __init__(self, criterion, geom_weight=1, level_nb=1):
        self.level_nb = level_nb
        if criterion == "l2":
            self.criterion = torch.nn.MSELoss(reduction="none")
        elif criterion == "l1":
            self.criterion = torch.nn.L1Loss(reduction="none")
        elif criterion == "ssim":
            raise NotImplementedError("ssim criterion needs kornia (out of scope, see DESIGN.md)")
        else:
            raise ValueError(f"{criterion} not in [l2, l1, ssim]")
        if level_nb != 1:
            raise NotImplementedError("level_nb > 1 needs kornia.ScalePyramid (out of scope, see DESIGN.md)")
        self.geom_weight = geom_weight

    def compute(self, inp, target, mask=None):
        """reference pyramidloss.py:56-62 (level_nb == 1 branch)."""
        diff = self.criterion(inp, target)
        losses = lossutils.batch_masked_mean_loss(diff, mask)
        return [inp], [target], losses, [diff], [mask]
