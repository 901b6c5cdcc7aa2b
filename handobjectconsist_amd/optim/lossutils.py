"""Per-sample masked mean -- same contract as meshreg/optim/lossutils.py:1-8 (pinned by
tests/golden/warp_misc.npz): mean of `dists` over the elements where `mask` is set, computed
separately for every batch entry; an entry whose mask is empty yields 0 (divides by 1)."""
import torch


def batch_masked_mean_loss(dists, mask):
    weights = mask.to(dists.dtype).flatten(1)
    numer = (weights * dists.flatten(1)).sum(1)
    count = weights.sum(1)
    return numer / torch.where(count == 0, torch.ones_like(count), count)
