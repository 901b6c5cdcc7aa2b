"""Which part of a step breaks hipGraph capture at a given size?  python scripts/graph_debug.py B SIZE STAGE
STAGE: prepare | data | consist | backward | step"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from handobjectconsist_amd.models.synthnet import SynthMeshRegNet
from handobjectconsist_amd.models.warpreg import WarpRegNet
from handobjectconsist_amd.netscripts import epochpassconsist as E

B, is_, stage = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = SynthMeshRegNet().to(dev).eval()
pre = WarpRegNet((is_, is_), model, lambda_consist=0.001, lambda_data=0.999, criterion="l1", gt_refs=True, progressive_steps=6,
                 use_backward=True, mano_faces=model.mano_layer.th_faces, pair_outputs="loss").to(dev)
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=5e-5, fused=True, capturable=True)
ld = E.SyntheticConsistLoader(B, is_, seed=3, device=dev, pool=1)
batches = ld.step_batches(0)
for _ in range(2):
    E.train_step(batches, pre, opt)
torch.cuda.synchronize()
pre.refresh_lambda_tensors()


def body():
    if stage == "step":
        return E.train_step(batches, pre, opt)[0]
    pre.prepare(batches, batch_encoder=True)
    if stage == "prepare":
        return None
    losses = [pre.forward(batches[0])[0].flatten()]
    if stage != "data":
        losses.append(pre.forward(batches[1])[0].flatten())
    loss = torch.stack(losses).sum()
    if stage == "backward":
        opt.zero_grad(set_to_none=True)
        loss.backward()
    return loss


g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = body()
print("captured", stage, flush=True)
g.replay()
torch.cuda.synchronize()
print("replayed", stage, None if out is None else float(out), flush=True)
