"""Does the hipGraph replay of SynthMeshRegNet.post_heads work and what does it buy?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from handobjectconsist_amd.models.synthnet import SynthMeshRegNet
from handobjectconsist_amd.models.warpreg import WarpRegNet
from handobjectconsist_amd.netscripts import epochpassconsist as E
dev = torch.device("cuda:0")
B, is_ = int(os.environ.get("B", 64)), int(os.environ.get("IS", 256))
res = {}
for graphs in (False, True):
    torch.manual_seed(0)
    model = SynthMeshRegNet().to(dev).eval()
    pre = WarpRegNet((is_, is_), model, lambda_consist=0.001, lambda_data=0.999, mano_faces=model.mano_layer.th_faces, pair_outputs="loss").to(dev)
    pre.step_count = 1000
    opt = torch.optim.Adam(model.parameters(), lr=5e-5)
    loader = E.SyntheticConsistLoader(B, is_, seed=0, device=dev, pool=2)
    if graphs:
        data, consist = loader.step_batches(0)
        model.enable_post_graphs(data["data"] + consist["data"])
    losses = []
    for i in range(5): losses.append(float(E.train_step(loader.step_batches(i), pre, opt)[0]))
    torch.cuda.synchronize()
    N = 10
    t0 = time.perf_counter()
    for i in range(N): E.train_step(loader.step_batches(i), pre, opt)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    res[graphs] = losses
    print(f"post graphs={graphs}: {1e3 * (t2 - t0) / N:.2f} ms/step   losses {['%.6f' % l for l in losses]}")
