"""aten / custom ops of one training step by device time and launch count (torch profiler), to locate the short-launch
tail.  GPU box: python scripts/step_ops.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from handobjectconsist_amd.models.synthnet import SynthMeshRegNet
from handobjectconsist_amd.models.warpreg import WarpRegNet
from handobjectconsist_amd.netscripts.epochpassconsist import SyntheticConsistLoader, train_step

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
B, is_ = 64, 256
model = SynthMeshRegNet().to(dev).eval()
pre = WarpRegNet((is_, is_), model, lambda_consist=0.001, lambda_data=0.999, criterion="l1", gt_refs=True, progressive_steps=1000,
                 use_backward=True, mano_faces=model.mano_layer.th_faces, pair_outputs="loss").to(dev)
pre.step_count = 1000
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=5e-5, fused=True)
loader = SyntheticConsistLoader(B, is_, seed=0, device=dev, pool=2)
for i in range(6):
    train_step(loader.step_batches(i), pre, opt)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    train_step(loader.step_batches(0), pre, opt)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages():
    dt = getattr(e, "self_device_time_total", None)
    if dt is None:
        dt = e.self_cuda_time_total
    if dt > 0:
        rows.append((dt, e.count, e.key))
rows.sort(reverse=True)
print("device us   calls  op")
for dt, n, k in rows[:70]:
    print(f"{dt:10.1f} {n:6d}  {k[:90]}")
