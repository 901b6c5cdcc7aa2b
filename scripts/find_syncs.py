"""List the CUDA-synchronising calls of one training step (torch.cuda.set_sync_debug_mode)."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from handobjectconsist_amd.models.synthnet import SynthMeshRegNet
from handobjectconsist_amd.models.warpreg import WarpRegNet
from handobjectconsist_amd.netscripts import epochpassconsist as E
dev = torch.device("cuda:0")
B, is_ = 8, 128
model = SynthMeshRegNet().to(dev).eval()
pre = WarpRegNet((is_, is_), model, lambda_consist=0.001, lambda_data=0.999, mano_faces=model.mano_layer.th_faces, pair_outputs="loss").to(dev)
pre.step_count = 1000
opt = torch.optim.Adam(model.parameters(), lr=5e-5)
loader = E.SyntheticConsistLoader(B, is_, seed=0, device=dev, pool=2)
for i in range(3): E.train_step(loader.step_batches(i), pre, opt)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    E.train_step(loader.step_batches(0), pre, opt)
torch.cuda.set_sync_debug_mode("default")
print("synchronising calls in one step:", len(w))

for x in w[:20]:
    print(" ", x.filename.replace(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "."), x.lineno, str(x.message)[:100])
