"""Coarse wall-clock breakdown of one step (with synchronisation between phases)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from handobjectconsist_amd.models.synthnet import SynthMeshRegNet
from handobjectconsist_amd.models.warpreg import WarpRegNet
from handobjectconsist_amd.netscripts.epochpassconsist import SyntheticConsistLoader, train_step

dev = torch.device("cuda:0")
B, is_ = 64, 256
model = SynthMeshRegNet().to(dev).eval()
pre = WarpRegNet((is_, is_), model, lambda_consist=0.001, lambda_data=0.999, mano_faces=model.mano_layer.th_faces, pair_outputs="loss").to(dev)
pre.step_count = 1000
opt = torch.optim.Adam(model.parameters(), lr=5e-5)
loader = SyntheticConsistLoader(B, is_, seed=0, device=dev, pool=2)
for i in range(5): train_step(loader.step_batches(i), pre, opt)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
acc = {}
def add(k, dt): acc[k] = acc.get(k, 0) + dt
N = 5
for i in range(N):
    data, consist = loader.step_batches(i)
    t0 = T(); feats = model.base_net(data["data"][0]["image"]); t1 = T(); add("encoder fwd (1 of 3 calls)", t1 - t0)
    t0 = T(); l1, _, _, _ = pre.forward(data); t1 = T(); add("data batch forward (enc + heads + MANO + losses)", t1 - t0)
    t0 = T(); l2, _, _, _ = pre.forward(consist); t1 = T(); add("consist batch forward (2x enc/heads/MANO + hot path fwd)", t1 - t0)
    opt.zero_grad(set_to_none=True)
    loss = torch.stack([l1.flatten(), l2.flatten()]).sum()
    t0 = T(); loss.backward(); t1 = T(); add("backward (everything)", t1 - t0)
    t0 = T(); opt.step(); t1 = T(); add("Adam step", t1 - t0)
    # heads + MANO alone
    with torch.no_grad():
        t0 = T(); base = model.mano_base(feats); pose, shape = model.pose_reg(base), model.shape_reg(base); v, j = model.mano_layer(pose, th_betas=shape); t1 = T()
    add("heads + MANO LBS fwd (1 of 3 calls)", t1 - t0)
for k, v in acc.items(): print(f"{k:60s} {v / N * 1e3:8.2f} ms")
