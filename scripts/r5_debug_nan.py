"""Which configuration of the graph-replayed step goes NaN?  python scripts/r5_debug_nan.py B size height dtype work steps"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from handobjectconsist_amd.models.synthnet import SynthMeshRegNet
from handobjectconsist_amd.models.warpreg import WarpRegNet
from handobjectconsist_amd.netscripts import epochpassconsist as E
from handobjectconsist_amd.warping import opticalflow

B, is_, ih_, dtype, work, steps, graph = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5] == "1", int(sys.argv[6]), sys.argv[7] == "1"
opticalflow.USE_SCATTER_WORK = work
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
torch.manual_seed(0)
model = SynthMeshRegNet().to(dev).eval().to(memory_format=torch.channels_last)
if dtype == "bf16":
    model.encoder_dtype = torch.bfloat16
pre = WarpRegNet((is_, ih_), model, lambda_consist=0.001, lambda_data=0.999, criterion="l1", gt_refs=True, progressive_steps=1000,
                 use_backward=True, mano_faces=model.mano_layer.th_faces, pair_outputs="loss").to(dev)
pre.step_count = 1000
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=5e-5, fused=True, capturable=True)
loader = E.SyntheticConsistLoader(B, is_, seed=0, device=dev, pool=2, image_height=ih_ if ih_ != is_ else None)
step = E.GraphedTrainStep(pre, opt, experimental=True, allow_autocast=True) if graph else (lambda b: E.train_step(b, pre, opt))
bad = None
for i in range(steps):
    try:
        loss, logs = step(loader.step_batches(i))
    except ValueError as e:
        bad = (i, "raised: " + str(e)[:40]); break
    torch.cuda.synchronize()
    nan = [k for k, v in logs.items() if torch.is_tensor(v) and not torch.isfinite(v).all()]
    if nan or not torch.isfinite(loss).all():
        bad = (i, nan, float(loss)); break
print(f"B={B} {is_}x{ih_} {dtype} work={work} graph={graph}: ", "NaN at step %s" % (bad,) if bad else f"{steps} steps finite, loss {float(loss):.6f}")
