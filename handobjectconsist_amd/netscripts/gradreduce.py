"""Bucketed gradient all-reduce for the batch-sharded trainer: one process per GPU, the only
exchange of a step is the mean of the trainable parameters' gradients over the ranks (RCCL over xGMI on
MI355X; gloo in the CPU tests).

Why not ``torch.nn.parallel.DistributedDataParallel``: a step of epochpassconsist.py:57-68 is SEVERAL forward
passes (data batch, both frames of the consist batch, here also the shared ``prepare`` pass) and ONE backward.
The wrapper re-arms its reducer and walks its inputs on every forward and post-processes every parameter's
gradient on its own (a copy / scale launch per parameter): measured on one MI355X, one rank, no byte
communicated: 30.1-30.7 ms per step against 26.6 ms without the wrapper, 430 against 352 launches
(profiles/r02_ddp_one_rank_overlap.txt).  Nothing of that is needed here:

* the model is NOT wrapped -- forwards cost what they cost without data parallelism;
* parameters are grouped, in reverse registration order (the order their gradients become ready in), into
  buckets of ``bucket_mb`` (xGMI is point-to-point: a bucket must keep a ring step per link well above its latency; 16 MB since
  round 5 -- four buckets for this trainer: one-rank cost 2.2 % with nine 8 MB buckets, 1.8 % with four or two,
  profiles/r05_reducer_bucket_sizes.txt);
  each bucket owns ONE flat buffer and per-parameter views of it with the parameter's own (dense) strides;
* a post-accumulate hook per parameter counts the bucket down; the hook of the LAST gradient of a bucket copies
  the bucket's gradients into the flat buffer with one multi-tensor copy, points ``p.grad`` at the views and
  issues ONE asynchronous all-reduce (sum) of the flat buffer -- on the collective's own stream, behind the
  copy, overlapping the rest of the backward pass (the encoder's, ~17 ms at B = 64);
* the mean over the ranks comes from scaling the LOSS by ``loss_scale`` = 1 / world size before ``backward()``
  (``train_step`` does it; exact for the power-of-two rank counts of a node) instead of from an averaging
  collective or a division pass over the gradients (RCCL's AVG is a pre-multiplied sum: an extra 47.9 MB
  scaling kernel per step even on one rank, profiles/r03_one_rank_reducer_overlap_v1.txt);
* ``finish()`` after ``backward()`` flushes buckets with parameters that received no gradient (zeros, so that
  every rank reduces the same buckets) and makes the compute stream wait for the collectives; the optimiser
  then reads the averaged gradients straight from the views;
* collectives are issued STRICTLY in bucket order on every rank: a bucket whose gradients are complete waits for
  its predecessors (a rank whose graph happens to finish bucket 3 before bucket 2 must not pair its all-reduce
  with another rank's bucket 2 -- RCCL matches collectives by issue order, not by buffer);
* one extra element behind the last bucket's gradients carries a per-step FLAG through the same all-reduce
  (``set_flag`` before ``backward()``, ``flag()`` after ``finish()``): the trainer's NaN guard becomes a
  decision every rank takes identically, at no extra collective.

All ranks must train the same parameter set.  A parameter that received no gradient on a rank contributes zeros
and ends up with a zero (not ``None``) gradient: under Adam without weight decay (trainmeshwarp.py's optimiser)
such a parameter does not move while its moments are zero, but unlike the single-process path its step counter
advances; ``check_optimizer`` (called by ``train_step``) refuses weight decay / momentum, under which it WOULD move.
The flag slot takes the last bucket's dtype (fp32 for this trainer; a 16-bit last bucket still carries 0 / non-zero
for up to 2048 ranks) and is re-zeroed by the last bucket's launch in steps that do not call ``set_flag``.

The same code runs for every world size, including 1 (no short cut: the one-rank run is how the cost of the
path is measured on a one-GPU box).
"""
import os

import torch
import torch.distributed as dist

_WAIT_EVERY_BUCKET = os.environ.get("HOC_REDUCER_WAIT_EVERY_BUCKET", "0") == "1"


class _Bucket:
    __slots__ = ("params", "flat", "views", "pending", "work", "launched")

    def __init__(self, params, extra=0):
        self.params = params
        total = sum(p.numel() for p in params)
        first = params[0]
        self.flat = torch.zeros(total + extra, dtype=first.dtype, device=first.device)
        self.views, off = [], 0
        for p in params:
            n = p.numel()
            chunk = self.flat[off:off + n]
            dense = p.is_contiguous() or _is_dense_permutation(p)
            self.views.append(chunk.as_strided(p.size(), p.stride()) if dense and n > 0 else chunk.view(p.size()))
            off += n
        self.pending, self.work, self.launched = len(params), None, False


def _is_dense_permutation(t):
    """True when ``t``'s strides are a permutation of a contiguous layout over exactly ``numel`` elements
    (channels-last convolution weights): a view with the same strides then fits a flat chunk of ``numel``."""
    expect = 1
    for size, stride in sorted(zip(t.size(), t.stride()), key=lambda s: (s[1], s[0])):
        if size == 1:
            continue
        if stride != expect:
            return False
        expect *= size
    return expect == t.numel()


class BucketedGradReducer:
    """``reducer = BucketedGradReducer(model.parameters())``; per step: forwards, ``(loss * reducer.loss_scale).backward()``,
    ``reducer.finish()``, ``optimizer.step()`` (with ``zero_grad`` anywhere before the next backward; ONE
    backward per ``finish()`` -- the step structure of epochpassconsist.py:57-68).  ``process_group=None`` = the
    default group."""

    def __init__(self, params, process_group=None, bucket_mb=16, broadcast_from=0):
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.backend = dist.get_backend(process_group)
        self.loss_scale = 1.0 / self.world  # multiply the loss by this before backward(): summed gradients = the mean
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("no trainable parameter")
        if broadcast_from is not None:
            self._broadcast(params, broadcast_from)
        cap = int(bucket_mb * (1 << 20))
        self.buckets, cur, cur_bytes, key = [], [], 0, None
        for p in reversed(params):
            k, nbytes = (p.dtype, p.device), p.numel() * p.element_size()
            if cur and (k != key or cur_bytes + nbytes > cap):
                self.buckets.append(_Bucket(cur))
                cur, cur_bytes = [], 0
            key = k
            cur.append(p)
            cur_bytes += nbytes
        self.buckets.append(_Bucket(cur, extra=1))
        self._flag = self.buckets[-1].flat[-1:]  # rides through the last bucket's all-reduce (sum over the ranks)
        self._next = 0  # index of the first bucket not yet issued in this step
        self._flag_set = False  # set_flag called since the last finish()
        self._handles = []
        for b in self.buckets:
            for p in b.params:
                self._handles.append(p.register_post_accumulate_grad_hook(self._make_hook(b)))
        self.grad_bytes = sum(b.flat.numel() * b.flat.element_size() for b in self.buckets)

    # ------------------------------------------------------------------ set-up
    def _broadcast(self, params, src):
        """Replicas start from rank ``src``'s parameters (one flat broadcast per dtype / device group)."""
        groups = {}
        for p in params:
            groups.setdefault((p.dtype, p.device), []).append(p)
        with torch.no_grad():
            for plist in groups.values():
                flat = torch.cat([p.detach().reshape(-1) if p.is_contiguous() else p.detach().contiguous().reshape(-1)
                                  for p in plist])
                dist.broadcast(flat, src=src, group=self.group)
                off = 0
                for p in plist:
                    n = p.numel()
                    p.copy_(flat[off:off + n].view(p.size()))  # logical order; strides of p are kept
                    off += n

    def _make_hook(self, bucket):
        def hook(_param):
            bucket.pending -= 1
            # strictly in bucket order: a complete bucket behind an incomplete one waits for it
            while self._next < len(self.buckets) and self.buckets[self._next].pending == 0:
                self._launch(self.buckets[self._next])
                self._next += 1
        return hook

    def set_flag(self, flag):
        """Before ``backward()``: this rank's flag of the step (0-dim or 1-element tensor or Python number, non-zero =
        raised).  ``flag()`` after ``finish()`` tells whether ANY rank raised it."""
        with torch.no_grad():
            if torch.is_tensor(flag):
                self._flag.copy_(flag.reshape(1).to(self._flag.dtype))
            else:
                self._flag.fill_(float(bool(flag)))
        self._flag_set = True

    def check_optimizer(self, optimizer):
        """A parameter without a gradient gets ZEROS here (every rank must reduce the same buckets), not ``None``: with
        weight decay or momentum the optimiser would move it, unlike the single-process path.  Refuse such settings."""
        for g in optimizer.param_groups:
            if g.get("weight_decay", 0) or g.get("momentum", 0):
                raise ValueError("BucketedGradReducer hands zero (not None) gradients to parameters without a gradient: "
                                 "weight_decay / momentum != 0 would move them; not supported")

    def flag(self):
        """After ``finish()``: 0-dim bool tensor on the parameters' device, identical on every rank."""
        return (self._flag != 0).reshape(()).clone()

    # ------------------------------------------------------------------ per step
    def _launch(self, b):
        with torch.no_grad():
            if b is self.buckets[-1] and not self._flag_set:
                self._flag.zero_()  # (a step without set_flag must not re-send the previous step's rank sum)
            have = [(v, p.grad) for v, p in zip(b.views, b.params)
                    if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
            if have:
                torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
            for v, p in zip(b.views, b.params):
                if p.grad is None:
                    v.zero_()  # a parameter without a gradient contributes zeros on this rank
                p.grad = v
        b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        b.launched = True

    def finish(self):
        """After ``backward()``: every bucket reduced, the current stream ordered behind the collectives,
        ``p.grad`` = the rank-summed gradient of the pre-scaled loss = the mean gradient (a view into the bucket's flat buffer)."""
        for b in self.buckets[self._next:]:
            self._launch(b)
        if self.backend == "nccl" and not _WAIT_EVERY_BUCKET:
            # RCCL runs a group's collectives on ONE stream in issue order: the last bucket's completion implies the others'.
            # One cross-stream wait instead of one per bucket (each is a barrier packet the compute queue stalls on: nine of
            # them were most of the 0.65 ms this path cost a one-rank step, profiles/r04_one_rank_reducer_vs_plain.json).
            # That order is ProcessGroupNCCL's implementation, not its contract: the buckets are issued in list order by
            # construction (`_launch` is only reached through `_next`), the last one issued is the last of the list -- asserted --
            # and HOC_REDUCER_WAIT_EVERY_BUCKET=1 restores one wait per bucket (errors / time-outs of the earlier buckets then
            # surface at their own wait instead of through the watchdog).
            assert all(b.launched and b.work is not None for b in self.buckets), "a bucket was never issued"
            self.buckets[-1].work.wait()
        else:
            for b in self.buckets:
                b.work.wait()
        for b in self.buckets:
            b.work, b.launched, b.pending = None, False, len(b.params)
        self._next = 0
        self._flag_set = False

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []
