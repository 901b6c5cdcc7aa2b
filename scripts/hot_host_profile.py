"""Where the HOST time of the eager render + warp hot path goes (cProfile over warpbranch.forward + backward, "loss" mode):
    python scripts/hot_host_profile.py [iterations]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from handobjectconsist_amd.models import warpbranch
from handobjectconsist_amd.models.synthnet import SynthMeshRegNet
from handobjectconsist_amd.models.warpreg import WarpRegNet
from handobjectconsist_amd.netscripts.epochpassconsist import SyntheticConsistLoader

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
model = SynthMeshRegNet().to(dev).eval()
pre = WarpRegNet((256, 256), model, lambda_consist=0.001, lambda_data=0.999, criterion="l1", gt_refs=True, progressive_steps=1000,
                 use_backward=True, mano_faces=model.mano_layer.th_faces, pair_outputs="loss").to(dev)
loader = SyntheticConsistLoader(64, 256, seed=0, device=dev, pool=1)
consist = loader.step_batches(0)[1]
fake = [{"recov_handverts3d": s_["_handverts3d"].clone().requires_grad_(True),
         "recov_objverts3d": s_["_objverts3d"].clone().requires_grad_(True)} for s_ in consist["data"]]


def hot():
    l, _ = warpbranch.forward(consist["data"], fake, pre.th_faces, pre.renderer, (256, 256), pre.criterion, gt_refs=True,
                              hand_ignore_faces=pre.hand_ignore_faces, use_backward=True, pair_outputs="loss")
    l.backward()


for _ in range(20):
    hot()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    hot()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative")
total = sum(v[3] for k, v in st.stats.items() if k[2] == "hot")
print(f"host time per pass: {total / n * 1e6:.0f} us")
st.print_stats(28)
