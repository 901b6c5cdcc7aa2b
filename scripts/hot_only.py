"""The render + warp hot path alone (warpbranch.forward + backward to the vertices, "loss" mode), N passes eager and N as a
hipGraph replay -- the workload of bench.py's hot_path leg without the trainer around it (no MIOpen solver search: starts in
seconds).  For rocprofv3 --kernel-trace (scripts/hot_kernels.sh) and quick A / B runs:
    python scripts/hot_only.py [--batch 64] [--image-size 256] [--image-height H] [--passes 30]
prints {"eager_ms": host-bound wall time per pass, "graph_ms": device time per pass}."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from handobjectconsist_amd.models import warpbranch
from handobjectconsist_amd.models.synthnet import SynthMeshRegNet
from handobjectconsist_amd.models.warpreg import WarpRegNet
from handobjectconsist_amd.netscripts.epochpassconsist import SyntheticConsistLoader

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--image-size", type=int, default=256)
ap.add_argument("--image-height", type=int, default=0)
ap.add_argument("--passes", type=int, default=30)
ap.add_argument("--no-graph", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
W, H = a.image_size, (a.image_height or a.image_size)
model = SynthMeshRegNet().to(dev).eval()
pre = WarpRegNet((W, H), model, lambda_consist=0.001, lambda_data=0.999, criterion="l1", gt_refs=True, progressive_steps=1000,
                 use_backward=True, mano_faces=model.mano_layer.th_faces, pair_outputs="loss").to(dev)
loader = SyntheticConsistLoader(a.batch, W, seed=0, device=dev, pool=1, image_height=H) if a.image_height else \
    SyntheticConsistLoader(a.batch, W, seed=0, device=dev, pool=1)
consist = loader.step_batches(0)[1]
fake = [{"recov_handverts3d": s_["_handverts3d"].clone().requires_grad_(True),
         "recov_objverts3d": s_["_objverts3d"].clone().requires_grad_(True)} for s_ in consist["data"]]
leaves = [v for r_ in fake for v in r_.values()]


def hot():
    l, _ = warpbranch.forward(consist["data"], fake, pre.th_faces, pre.renderer, (W, H), pre.criterion, gt_refs=True,
                              hand_ignore_faces=pre.hand_ignore_faces, use_backward=True, pair_outputs="loss")
    return torch.autograd.grad(l, leaves, allow_unused=True)


for _ in range(5):
    hot()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.passes):
    hot()
torch.cuda.synchronize()
out = {"eager_ms": round((time.perf_counter() - t0) / a.passes * 1e3, 4)}
if not a.no_graph:
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            hot()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):  # (the stream of the passes above: the pair step's plan is keyed by stream)
        hot()
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.passes):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    out["graph_ms"] = round(e0.elapsed_time(e1) / a.passes, 4)
print(json.dumps(out))
