"""EXPERIMENT, not product code: ``train_step`` captured once per batch set into a hipGraph and replayed.

Moved here from handobjectconsist_amd/netscripts/epochpassconsist.py in round 6: replayed steps return garbage convolution
weight gradients now and then (round 5's twin-model test), which is not root-caused, and nothing known-wrong ships in the
package.  The product's step is the eager ``train_step``.  Used by scripts/graph_step_bisect.py (the twin comparison under
MIOpen solver restrictions) and scripts/host_timeline.py (host issue time of a replayed step, timing only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from handobjectconsist_amd.netscripts.epochpassconsist import _device_guarded, raise_pending_nan, train_step  # noqa: E402


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

    def __init__(self, premodel, optimizer, check_nan=True, max_graphs=8, experimental=False, allow_autocast=False, warm_iters=1):
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
        self.warm_iters = max(1, int(warm_iters))  # eager side-stream calls of a batch set before its capture
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
            self._entries[key] = {"batches": batches, "graph": None, "warm": 0}
            entry = self._entries[key]
        if entry["graph"] is None and entry["warm"] < self.warm_iters:
            entry["warm"] += 1
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
