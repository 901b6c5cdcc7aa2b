"""Twin-model comparison of a hipGraph-replayed train step against the eager step, learning rate 0 (round 6, time-boxed).

    python scripts/graph_step_bisect.py [--replays 50] [--batch 4] [--size 128] [--warm 3]

One model is stepped eagerly, its twin (same seed) through scripts/graph_step_experiment.GraphedTrainStep; after every step the
gradients of all parameters are compared.  Prints one line per bad step (relative difference above 1e-3) naming the worst
parameters, and a summary line `bad=<n>/<replays>`.  Run under different MIOpen settings (environment) by the caller:
scripts/graph_step_bisect.sh."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from graph_step_experiment import GraphedTrainStep  # noqa: E402

from handobjectconsist_amd.models.synthnet import SynthMeshRegNet  # noqa: E402
from handobjectconsist_amd.models.warpreg import WarpRegNet  # noqa: E402
from handobjectconsist_amd.netscripts.epochpassconsist import SyntheticConsistLoader, train_step  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--replays", type=int, default=50)
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--size", type=int, default=128)
ap.add_argument("--warm", type=int, default=3, help="eager side-stream iterations of a batch set before its capture")
ap.add_argument("--benchmark", type=int, default=0, help="torch.backends.cudnn.benchmark (MIOpen's measured solver search)")
args = ap.parse_args()
torch.backends.cudnn.benchmark = bool(args.benchmark)
dev = torch.device("cuda", 0)


def build(seed):
    torch.manual_seed(seed)
    model = SynthMeshRegNet().to(dev)
    model.eval()
    pre = WarpRegNet((args.size, args.size), model, lambda_consist=0.001, lambda_data=0.999, criterion="l1", gt_refs=True,
                     progressive_steps=6, use_backward=True, mano_faces=model.mano_layer.th_faces, pair_outputs="loss").to(dev)
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=0.0, fused=True, capturable=True)
    loader = SyntheticConsistLoader(args.batch, args.size, seed=3, device=dev, pool=2)
    return model, pre, opt, loader


model_e, pre_e, opt_e, loader_e = build(11)
model_g, pre_g, opt_g, loader_g = build(11)
names = [n for n, p in model_g.named_parameters() if p.requires_grad]
step_g = GraphedTrainStep(pre_g, opt_g, check_nan=False, experimental=True, warm_iters=args.warm)
bad = 0
sets = len(loader_g.batches)
for i in range(args.replays + sets * (args.warm + 1)):
    train_step(loader_e.step_batches(i), pre_e, opt_e, check_nan=False)
    ge = [p.grad.detach().clone() for g in opt_e.param_groups for p in g["params"]]
    before = step_g.replays
    step_g(loader_g.step_batches(i))
    if step_g.replays == before:
        continue  # (an eager warm-up call of this batch set)
    gg = step_g.last_grads
    num = torch.stack([(a.double() - b.double()).norm() for a, b in zip(ge, gg)])
    den = torch.cat([a.flatten() for a in ge]).double().norm()
    rel = float(num.norm() / den)
    if not rel < 1e-3:
        bad += 1
        order = torch.argsort(num, descending=True)[:4].tolist()
        print("step %d (replay %d): relative difference %.3e; worst: %s" % (
            i, step_g.replays, rel, ", ".join("%s %.2e" % (names[k], float(num[k] / den)) for k in order)), flush=True)
print("bad=%d/%d replays  (B=%d, %dx%d, warm=%d, cudnn.benchmark=%d, MIOPEN env: %s)" % (
    bad, step_g.replays, args.batch, args.size, args.size, args.warm, args.benchmark,
    " ".join("%s=%s" % kv for kv in sorted(os.environ.items()) if kv[0].startswith("MIOPEN")) or "-"), flush=True)
