"""Host-side time stamps of the training step's phases against the device time of the step (the bench workload): where does
the host issue ahead of the device and where does the device wait for it?
    python scripts/host_timeline.py [--steps N] [--batch B] [--image-size S] [--image-height H] [--encoder-dtype f32|bf16]
Prints the eager step (device time between two events, host time to issue it, host time until the device is done) and the
same step replayed from a hipGraph (netscripts/epochpassconsist.GraphedTrainStep): one launch per step."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from graph_step_experiment import GraphedTrainStep  # noqa: E402  (experiment, not product: see that file)

from handobjectconsist_amd.models.synthnet import SynthMeshRegNet
from handobjectconsist_amd.models.warpreg import WarpRegNet
from handobjectconsist_amd.netscripts import epochpassconsist as E

import argparse

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--image-size", type=int, default=256)
ap.add_argument("--image-height", type=int, default=None)
ap.add_argument("--encoder-dtype", default="f32")
ap.add_argument("--no-graph", action="store_true", help="the eager step only")
args = ap.parse_args()
dev = torch.device("cuda:0")
n, B, is_, ih_ = args.steps, args.batch, args.image_size, args.image_height or args.image_size
torch.backends.cudnn.benchmark = False  # (the replayed step below refuses MIOpen's solver search; eager and replay on the same solvers)
model = SynthMeshRegNet().to(dev).eval()
if args.encoder_dtype == "bf16":
    model.encoder_dtype = torch.bfloat16
if os.environ.get("HOC_CHANNELS_LAST", "1") == "1":
    model = model.to(memory_format=torch.channels_last)
pre = WarpRegNet((is_, ih_), model, lambda_consist=0.001, lambda_data=0.999, criterion="l1", gt_refs=True, progressive_steps=1000,
                 use_backward=True, mano_faces=model.mano_layer.th_faces, pair_outputs="loss").to(dev)
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=5e-5, fused=True, capturable=True)
loader = E.SyntheticConsistLoader(B, is_, seed=0, device=dev, pool=2, image_height=args.image_height)

stamps = []
real_forward = pre.forward
real_backward = torch.Tensor.backward


def fwd(batch):
    t0 = time.perf_counter()
    out = real_forward(batch)
    stamps.append(("forward", t0, time.perf_counter()))
    return out


def bwd(self, *a, **k):
    t0 = time.perf_counter()
    r = real_backward(self, *a, **k)
    stamps.append(("backward", t0, time.perf_counter()))
    return r


pre.forward = fwd
torch.Tensor.backward = bwd
for i in range(8):
    E.train_step(loader.step_batches(i), pre, opt)
torch.cuda.synchronize()
del stamps[:]
rows = []
for i in range(n):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    batches = loader.step_batches(i)
    t0 = time.perf_counter()
    e0.record()
    E.train_step(batches, pre, opt)
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ph = {}
    for name, a, b in stamps:
        ph.setdefault(name, []).append(((a - t0) * 1e3, (b - t0) * 1e3))
    del stamps[:]
    rows.append((e0.elapsed_time(e1), (t1 - t0) * 1e3, (t2 - t0) * 1e3, ph))
import statistics as st
print("B=%d %dx%d %s" % (B, is_, ih_, args.encoder_dtype))
print("eager step (ms): device %.2f   host issue %.2f   host until device done %.2f   host issue / device %.2f" % (
    st.median(r[0] for r in rows), st.median(r[1] for r in rows), st.median(r[2] for r in rows),
    st.median(r[1] for r in rows) / st.median(r[0] for r in rows)))
r = rows[len(rows) // 2]
for name, spans in r[3].items():
    print("  %-9s" % name, "  ".join("%.2f-%.2f" % s for s in spans))
if args.no_graph:
    sys.exit(0)
# the same step as ONE graph launch
pre.forward = real_forward
torch.Tensor.backward = real_backward
step = GraphedTrainStep(pre, opt, experimental=True, allow_autocast=True)  # (bf16: measured here, not used by bench.py -- see its note)
for i in range(6):
    step(loader.step_batches(i))
torch.cuda.synchronize()
rows = []
for i in range(n):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    step(loader.step_batches(i))
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    rows.append((e0.elapsed_time(e1), (t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3))
E.raise_pending_nan(opt)
print("graph replay (ms): device %.2f   host issue %.2f   host until device done %.2f" % (
    st.median(r[0] for r in rows), st.median(r[1] for r in rows), st.median(r[2] for r in rows)))
