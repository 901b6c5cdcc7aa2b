"""GraphedTrainStep at a given size, one model:  python scripts/graph_debug2.py B SIZE [two]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faulthandler; faulthandler.enable()
import torch

import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from graph_step_experiment import GraphedTrainStep  # noqa: E402  (experiment, not product: see that file)
from handobjectconsist_amd.models.synthnet import SynthMeshRegNet
from handobjectconsist_amd.models.warpreg import WarpRegNet
from handobjectconsist_amd.netscripts import epochpassconsist as E

B, is_ = int(sys.argv[1]), int(sys.argv[2])
two = len(sys.argv) > 3
dev = torch.device("cuda:0")


def build(capturable):
    torch.manual_seed(0)
    model = SynthMeshRegNet().to(dev).eval()
    pre = WarpRegNet((is_, is_), model, lambda_consist=0.001, lambda_data=0.999, criterion="l1", gt_refs=True, progressive_steps=6,
                     use_backward=True, mano_faces=model.mano_layer.th_faces, pair_outputs="loss").to(dev)
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=5e-5, fused=True, capturable=capturable)
    ld = E.SyntheticConsistLoader(B, is_, seed=3, device=dev, pool=2)
    return model, pre, opt, ld


if two:
    me, pe, oe, le = build(False)
mg, pg, og, lg = build(True)
step = GraphedTrainStep(pg, og, experimental=True)
for i in range(6):
    if two:
        E.train_step(le.step_batches(i), pe, oe)
    l, _ = step(lg.step_batches(i))
    print(i, float(l), step.replays, flush=True)
E.raise_pending_nan(og)
print("ok", flush=True)
