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


class GraphedTrainStep:
    """``train_step`` captured ONCE per set of device-resident batches into a hipGraph and replayed ("host
    overhead dominates once kernels are fast").  A step of the metric workload is 333 launches; between the end of the
    encoder's forward and the first large kernel of its backward lie ~170 launches of a few microseconds (heads, MANO,
    losses, render + warp, the start of backward) that the autograd engine cannot issue as fast as the device retires them:
    1.0 ms of a 27.3 ms step is idle there (profiles/r04_step_sequence.txt), and at the reference's default batch size
    (B = 8, trainmeshwarp.py:372) the host binds the whole step.  One graph launch per step removes the host from it.

    What the capture holds: ``prepare`` + the forwards of the step's batches + ``zero_grad`` + ``backward`` + the fused
    optimiser update with its device-side NaN guard (``found_inf``) -- i.e. ``train_step`` itself, called under
    ``torch.cuda.graph``.  What stays outside: ``raise_pending_nan`` (the host reads step k's flag when step k + 1 starts,
    as in the eager path), the lambda ramp (``WarpRegNet.refresh_lambda_tensors``: a device tensor refreshed before the
    replay, nothing once the ramp is over) and the step counter.  The graph READS THE BATCH TENSORS IN PLACE: a batch set
    is identified by its dict objects, whose tensors must keep their storage and may be refilled in place between steps (a
    frame pipeline writing into fixed buffers -- ``mr_frames_to_batch`` -- or a device-resident pool as
    ``SyntheticConsistLoader``).  The first call with a new batch set runs eagerly (solver searches, TunableOp, the tile-list
    guess), the second captures and replays, later ones replay.

    **EXPERIMENTAL -- NOT RELIABLE, measurements only** (round 5).  A twin-model test (one model stepped eagerly, its twin through
    the replay, learning rate 0) shows that a replayed step NOW AND THEN returns garbage in the weight gradients of the trunk's
    convolutions -- 2e5 x the gradient's norm in conv1 / layer1, losses unchanged: 1 step in 8 at 128 x 128 with MIOpen's
    solver search on, step 8 of 9 in another run with it off; eager steps never.  Not root-caused (MIOpen's split-K
    weight-gradient kernels under capture are the suspects); the NaN head losses of replayed bf16 runs came from it.  The
    constructor therefore refuses to build unless ``experimental=True``; ``train_step`` (eager) is the product's step, and the
    host never binds it by the 0.8 rule (issue / device time 0.21 at the metric config, 0.65 at B = 8, 0.33 at config 5:
    profiles/r05_host_timeline.txt).

    Requirements: a CUDA optimiser built with ``capturable=True`` (stock fused Adam), no ``reducer`` (no collective has run
    inside a capture on hardware here), ``check_nan`` handled on the device.  A premodel that goes through this class should not
    be stepped eagerly on another stream in between (its AccumulateGrad nodes are bound to the stream of their first backward
    pass)."""

    def __init__(self, premodel, optimizer, check_nan=True, max_graphs=8, experimental=False, allow_autocast=False):
        if not experimental:
            raise ValueError("GraphedTrainStep is experimental: replayed steps return garbage convolution weight gradients now "
                             "and then (see the class docstring); pass experimental=True for timing measurements only")
        if not _device_guarded(optimizer) or not all(g.get("capturable", False) for g in optimizer.param_groups):
            raise ValueError("GraphedTrainStep needs a fused optimiser built with capturable=True")
        enc_dtype = getattr(getattr(premodel, "model", None), "encoder_dtype", None)
        if enc_dtype not in (None, torch.float32) and not allow_autocast:
            raise ValueError("GraphedTrainStep is validated for an fp32 trunk only (see the class docstring); "
                             "pass allow_autocast=True to capture an autocast step anyway")
        self.premodel, self.optimizer, self.check_nan, self.max_graphs = premodel, optimizer, check_nan, max_graphs
        self._entries = {}
        self.replays = 0
        self.last_grads = None  # gradient tensors of the last replayed step, in the optimiser's parameter order
        # The eager first call of a batch set and the capture run on ONE side stream: autograd binds a parameter's
        # AccumulateGrad node to the stream it is created on, and a node left over from a default-stream backward inside a
        # capture on another stream breaks the capture (PyTorch's whole-network capture recipe warms up on a side stream
        # for that reason; on ROCm the broken capture ends in a segmentation fault in hipStreamEndCapture).
        self._stream = torch.cuda.Stream()

    def __call__(self, batches):
        key = tuple(id(b) for b in batches)
        entry = self._entries.get(key)
        if self.check_nan:
            raise_pending_nan(self.optimizer)
        if entry is None:
            if len(self._entries) >= self.max_graphs:
                raise RuntimeError("GraphedTrainStep: more batch sets than max_graphs -- refill the batch tensors in place "
                                   "instead of handing over new ones")
            self._entries[key] = {"batches": batches, "graph": None}
            self.premodel.refresh_lambda_tensors()  # (the eager step reads the same device tensor the captures will)
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                out = train_step(batches, self.premodel, self.optimizer, check_nan=self.check_nan)
            torch.cuda.current_stream().wait_stream(self._stream)
            return out
        pm = self.premodel
        pm.refresh_lambda_tensors()
        if entry["graph"] is None:
            self._capture(entry)
        entry["graph"].replay()
        self.replays += 1
        self.last_grads = entry["grads"]
        pm.step_count += entry["consist_batches"]
        if self.check_nan:
            self.optimizer._hoc_pending_nan = entry["nan_flag"]
        return entry["loss"], entry["logs"]

    def _capture(self, entry):
        pm, opt = self.premodel, self.optimizer
        tunable = getattr(torch.cuda, "tunable", None)
        was_tuning = bool(tunable and tunable.is_enabled() and tunable.tuning_is_enabled())
        if was_tuning:
            tunable.tuning_enable(False)  # (every GEMM shape of the step was tuned by the eager call; no timing runs in a capture)
        count0 = pm.step_count
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph, stream=self._stream):
                loss, logs = train_step(entry["batches"], pm, opt, check_nan=self.check_nan)
                flag = getattr(opt, "_hoc_pending_nan", None)
        finally:
            if was_tuning:
                tunable.tuning_enable(True)
        opt._hoc_pending_nan = None
        # (the gradient tensors the replay writes: `p.grad` points at them until somebody else resets the gradients)
        grads = [p.grad for g in opt.param_groups for p in g["params"]]
        entry.update(graph=graph, loss=loss, logs=logs, nan_flag=flag, consist_batches=pm.step_count - count0, grads=grads)
        pm.step_count = count0  # (nothing ran yet: the replay that follows is the step)


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
