"""Steady-state torch.profiler breakdown of one trainmeshwarp step (after MIOpen's find phase)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from handobjectconsist_amd.models.synthnet import SynthMeshRegNet
from handobjectconsist_amd.models.warpreg import WarpRegNet
from handobjectconsist_amd.netscripts.epochpassconsist import SyntheticConsistLoader, train_step

dev = torch.device("cuda:0")
B, is_ = 64, 256
model = SynthMeshRegNet().to(dev).eval()
pre = WarpRegNet((is_, is_), model, lambda_consist=0.001, lambda_data=0.999, mano_faces=model.mano_layer.th_faces,
                 pair_outputs="loss").to(dev)
pre.step_count = 1000
opt = torch.optim.Adam(model.parameters(), lr=5e-5)
loader = SyntheticConsistLoader(B, is_, seed=0, device=dev, pool=2)
for i in range(5):
    train_step(loader.step_batches(i), pre, opt)
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for i in range(N):
        train_step(loader.step_batches(i), pre, opt)
    torch.cuda.synchronize()
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in ka)
print(f"device time per step: {tot / N / 1e3:.2f} ms")
for e in rows[:45]:
    print(f"{e.key[:70]:70s} n={e.count // N:5d}  {e.self_device_time_total / N / 1e3:9.3f} ms/step")
