"""Launches and device time of the parameter-free code behind the heads (MANO kernels + PyTorch glue), 3 x B = 192 samples."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from handobjectconsist_amd.models.synthnet import SynthMeshRegNet
dev = torch.device("cuda:0")
B = 192
model = SynthMeshRegNet().to(dev).eval()
pose = (0.3 * torch.randn(B, 18, device=dev)).requires_grad_(True)
shape = torch.randn(B, 10, device=dev).requires_grad_(True)
st = torch.randn(B, 3, device=dev).requires_grad_(True)
so = torch.randn(B, 6, device=dev).requires_grad_(True)
K = torch.tensor([[350.0, 0, 128], [0, 350.0, 128], [0, 0, 1]], device=dev).repeat(B, 1, 1)
can = torch.randn(B, 1002, 3, device=dev) * 0.05
def run():
    out = model.post_heads(pose, shape, st, so, K, can, input_res=(256, 256))
    sum(o.sum() for o in out).backward()
for _ in range(3): run()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    run(); torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
tim = collections.Counter(); cnt = collections.Counter()
for e in evs:
    k = "mr:: (MANO)" if "mr::" in e.name else e.name[:60]
    cnt[k] += 1; tim[k] += e.device_time
print("launches", len(evs), "device us", round(sum(tim.values()), 1), "of which MANO kernels", round(tim["mr:: (MANO)"], 1), "us in", cnt["mr:: (MANO)"])
