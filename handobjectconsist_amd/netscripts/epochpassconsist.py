"""Training loop over alternating data / consist batches -- counterpart of the hot loop of
meshreg/netscripts/epochpassconsist.py:56-68 (metric meters, evaluators and figure dumps are
out of scope).  Loss is accumulated over ``loader_nb`` consecutive batches, then ONE
``zero_grad / backward / step``: that is the "iteration" of the headline metric.
"""
import os

import numpy as np
import torch

from handobjectconsist_amd.utils import synth


# BATCH_POST: the regression heads and the parameter-free code behind them (MANO LBS, camera recovery,
# projections) run ONCE over the frames of a step instead of once per frame (WarpRegNet.prepare).
# Per-sample operations only: same values.
# BATCH_ENCODER: also ONE encoder pass over all frames of the step (SURVEY Q16 / 8f "f2": legal because the
# BatchNorm statistics are frozen -- every layer then acts per image; WarpRegNet.prepare falls back to one
# pass per frame for a model in training mode).  Measured on MI355X at B=64, 256x256: ResNet-18 forward +
# backward over 3 x 64 images 41.8 ms, over 1 x 192 images 40.2 ms, and 124 gradient-accumulation launches
# fewer; whole step 43.45 -> 42.0 ms.  (An earlier measurement had shown the opposite -- 52.7 -> 57.9 ms --
# while the step still contained hidden host->device synchronisations; scripts/cpu_floor.py: the launch-bound
# floor of a step drops from 27 ms to 17 ms.)  HOC_BATCH_ENCODER=0 / HOC_BATCH_POST=0 select the reference's
# per-frame structure.
BATCH_POST = os.environ.get("HOC_BATCH_POST", "1") != "0"
BATCH_ENCODER = os.environ.get("HOC_BATCH_ENCODER", "1") != "0"


def _device_guarded(optimizer):
    """True for optimisers whose step can be switched off ON THE DEVICE by a flag tensor (stock PyTorch's fused
    Adam / AdamW / SGD: the `found_inf` operand GradScaler uses)."""
    return bool(getattr(optimizer, "_step_supports_amp_scaling", False))


def raise_pending_nan(optimizer):
    """Raise for a NaN loss flagged by an earlier ``train_step`` (see there)."""
    pending = getattr(optimizer, "_hoc_pending_nan", None)
    if pending is not None:
        optimizer._hoc_pending_nan = None
        if bool(pending):
            raise ValueError("Loss became nan! (the step's parameter update was skipped on the device)")


def train_step(batches, premodel, optimizer, check_nan=True, reducer=None):
    """One optimiser step over `loader_nb = len(batches)` batches (epochpassconsist.py:57-68).  With
    ``check_nan`` a NaN loss never reaches the parameters or the optimiser state, as in the reference (:61-63),
    and raises ``ValueError``.  The reference reads the loss on the host before ``backward`` -- a pipeline drain
    per step (measured here: 1.3 ms with the check before ``backward``, 0.7 ms before ``step``).  Here the flag
    stays on the device: it switches the fused optimiser's update off (its ``found_inf`` operand), and the host
    looks at it when the NEXT step starts, when it has long been computed.

    Contract for callers that drive ``train_step`` themselves (``epoch_pass`` does this for its callers): the
    ``ValueError`` of step k is raised by the call for step k + 1, so after the LAST step call
    ``raise_pending_nan(optimizer)`` -- otherwise a NaN in the last step goes unreported (its update was skipped, but
    the returned / logged loss of that step is NaN).  An ``optimizer.grad_scale`` / ``found_inf`` a caller has set
    (AMP's GradScaler protocol) is put back after the step.  Optimisers without the ``found_inf`` operand get the
    synchronous check before ``step`` (one host read per step, as in the reference).
    ``reducer`` (data-parallel runs): a ``gradreduce.BucketedGradReducer`` over the model's parameters; its
    all-reduces are issued from inside ``backward`` and joined before the optimiser reads the gradients.  The NaN
    flag is then GLOBAL: it rides through the last bucket's all-reduce, so a NaN loss on any rank skips the update
    and raises on every rank in the same step (the replicas stay identical and no rank is left waiting in a
    collective)."""
    if check_nan:
        raise_pending_nan(optimizer)
    losses, logs = [], {}
    if (BATCH_POST or BATCH_ENCODER) and hasattr(premodel, "prepare"):
        premodel.prepare(batches, batch_encoder=BATCH_ENCODER)
    try:
        for batch in batches:
            loss, all_losses, _results, _pair_results = premodel.forward(batch)
            losses.append(loss.flatten())
            # (detached: a log entry that kept its grad_fn would keep this step's autograd graph -- and the parameters'
            # AccumulateGrad nodes, bound to the stream they were created on -- alive in the caller's hands)
            logs.update({k: (v.detach() if torch.is_tensor(v) else v) for k, v in all_losses.items() if v is not None})
    finally:
        for batch in batches:  # the prepared tensors belong to this step's autograd graph
            for sample in batch["data"]:
                sample.pop("_features", None)
                sample.pop("_post", None)
    loss = torch.stack(losses).sum()
    nan_flag = torch.isnan(loss.detach()) if check_nan else None
    optimizer.zero_grad(set_to_none=True)
    if loss.requires_grad:
        if reducer is not None and getattr(optimizer, "_hoc_reducer_checked", None) is not reducer:
            reducer.check_optimizer(optimizer)  # (once per optimiser / reducer pair)
            optimizer._hoc_reducer_checked = reducer
        if reducer is not None and check_nan:
            reducer.set_flag(nan_flag)  # travels with the last gradient bucket
        if reducer is not None and reducer.loss_scale != 1.0:
            (loss * reducer.loss_scale).backward()  # summed over the ranks: the mean gradient
        else:
            loss.backward()
        if reducer is not None:
            reducer.finish()
            if check_nan:
                # the gradients are rank-summed: ONE rank's NaN loss is every rank's NaN gradient.  The guard must
                # therefore be the same decision everywhere -- any rank's flag skips the update (and raises) on all
                nan_flag = reducer.flag()
        if check_nan and _device_guarded(optimizer):
            had = {k: optimizer.__dict__[k] for k in ("grad_scale", "found_inf") if k in optimizer.__dict__}
            optimizer.grad_scale, optimizer.found_inf = None, nan_flag.to(torch.float32).reshape(())
            try:
                optimizer.step()
            finally:
                del optimizer.grad_scale, optimizer.found_inf
                optimizer.__dict__.update(had)
            optimizer._hoc_pending_nan = nan_flag
        else:
            if check_nan and bool(nan_flag):
                raise ValueError("Loss became nan!")
            optimizer.step()
    elif check_nan and bool(nan_flag):
        raise ValueError("Loss became nan!")
    return loss.detach(), logs


def epoch_pass(loader, premodel, optimizer, loader_nb=2, check_nan=True, reducer=None):
    """loader yields {"data": [sample, ...], "supervision": "data" | "consist"} dicts."""
    pending, history = [], []
    for batch in loader:
        pending.append(batch)
        if len(pending) == loader_nb:
            loss, _ = train_step(pending, premodel, optimizer, check_nan=check_nan, reducer=reducer)
            history.append(loss)
            pending = []
    if check_nan:
        raise_pending_nan(optimizer)
    return history


class SyntheticConsistLoader:
    """Device-resident synthetic stand-in for ConcatLoader(strong loader, consist loader)
    (concatloader.py:4-30; trainmeshwarp.py:97-156): alternates one "data" batch (one frame,
    fully supervised) and one "consist" batch (a frame pair: unannotated frame first, annotated
    reference second -- warpbranch compares everything to samples[0], GT replaces samples[1:])."""

    def __init__(self, batch_size, image_size=256, steps=1, seed=0, device="cuda", pool=2, image_height=None):
        self.steps, self.device = steps, torch.device(device)
        self.batches = []
        t = lambda a: torch.from_numpy(a).to(self.device)
        ov, _ = synth.object_template()
        for k in range(pool):
            s = synth.random_scene(batch_size, seed=seed * 1000 + k, image_size=image_size)
            # non-square frames (image_height < image_size): the scene is laid out for the square raster of the
            # longer side, the images are its top rows (the render is cropped to the top-left H x W)
            im_ref, im, jm_ref, jm = synth.random_images(batch_size, image_height or image_size, image_size, seed * 1000 + k)
            canverts = t(ov[None].repeat(batch_size, 0).copy())
            # the three frames of a step in ONE buffer, in the order the step consumes them (data frame, unannotated
            # frame, annotated reference), as a GPU-side frame pipeline (mr_frames_to_batch) lays them out: the
            # encoder's single pass over the step's frames then needs no concatenation copy
            frames = t(np.concatenate([im_ref, im, im_ref]))
            step_images = {"data": frames[:batch_size], "unannotated": frames[batch_size:2 * batch_size],
                           "reference": frames[2 * batch_size:]}

            def sample(img, jmask, hand, obj, K, supervised):
                d = {"image": img if torch.is_tensor(img) else t(img), "jittermask": t(jmask), "camintr": t(K), "objcanverts": canverts,
                     "objfaces": t(s["obj_faces"][None].repeat(batch_size, 0).copy()),
                     # geometry of the frame, NOT read by the model (kernel-only benchmarks use it)
                     "_handverts3d": t(hand), "_objverts3d": t(obj)}
                if supervised:
                    d.update({"handverts3d": t(hand), "objverts3d": t(obj),
                              "joints3d": t(hand[:, :21].copy())})
                return d

            data = {"data": [sample(step_images["data"], jm_ref, s["hand_verts2"], s["obj_verts2"], s["K2"], True)],
                    "supervision": "data"}
            consist = {"data": [sample(step_images["unannotated"], jm, s["hand_verts1"], s["obj_verts1"], s["K1"], False),
                                sample(step_images["reference"], jm_ref, s["hand_verts2"], s["obj_verts2"], s["K2"], True)],
                       "supervision": "consist"}
            self.batches.append((data, consist))

    def __iter__(self):
        for i in range(self.steps):
            data, consist = self.batches[i % len(self.batches)]
            yield data
            yield consist

    def step_batches(self, i):
        return list(self.batches[i % len(self.batches)])
