"""Kernel launches of the hot path, by kernel name (torch profiler)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from handobjectconsist_amd.models.synthnet import SynthMeshRegNet
from handobjectconsist_amd.models.warpreg import WarpRegNet
from handobjectconsist_amd.models import warpbranch
from handobjectconsist_amd.netscripts.epochpassconsist import SyntheticConsistLoader
dev = torch.device("cuda:0")
B, is_ = 64, 256
model = SynthMeshRegNet().to(dev).eval()
pre = WarpRegNet((is_, is_), model, lambda_consist=0.001, lambda_data=0.999, mano_faces=model.mano_layer.th_faces, pair_outputs="loss").to(dev)
loader = SyntheticConsistLoader(B, is_, seed=0, device=dev, pool=1)
_, consist = loader.step_batches(0)
def fake_results():
    out = []
    for s in consist["data"]:
        out.append({"recov_handverts3d": s["_handverts3d"].clone().requires_grad_(True), "recov_objverts3d": s["_objverts3d"].clone().requires_grad_(True)})
    return out
def hot():
    res = fake_results()
    l, _ = warpbranch.forward(consist["data"], res, pre.th_faces, pre.renderer, (is_, is_), pre.criterion, gt_refs=True,
                              hand_ignore_faces=pre.hand_ignore_faces, use_backward=True, pair_outputs="loss")
    l.backward()
for _ in range(3): hot()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    hot()
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
cnt, tim = collections.Counter(), collections.Counter()
for e in evs:
    cnt[e.name[:70]] += 1; tim[e.name[:70]] += e.device_time if hasattr(e, "device_time") else e.cuda_time
print("launches", len(evs), "total device us", sum(tim.values()))
for k, v in sorted(tim.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{cnt[k]:4d} x {v:9.1f} us  {k}")
