"""Eager vs graph-replayed gradients, parameter by parameter (the test's setting).  python scripts/r5_graph_grad_debug.py [lambda_consist]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_gpu_graph_step import _build
from handobjectconsist_amd.netscripts.epochpassconsist import GraphedTrainStep, train_step
dev = torch.device("cuda:0")
model, pre, opt, loader = _build(dev, 4, 128, 11, True, lr=0.0)
if len(sys.argv) > 1:
    pre.lambda_consist = float(sys.argv[1])
names = [n for n, p in model.named_parameters() if p.requires_grad]
params = [p for g in opt.param_groups for p in g["params"]]
step_g = GraphedTrainStep(pre, opt, experimental=True)
def grads(): return [p.grad.detach().clone() for p in params]
for i in range(6):
    pre.step_count = i
    train_step(loader.step_batches(i), pre, opt); e1 = grads()
    pre.step_count = i
    train_step(loader.step_batches(i), pre, opt); e2 = grads()
    pre.step_count = i
    lg, logs = step_g(loader.step_batches(i))
    g = [x.detach().clone() for x in (step_g.last_grads if i >= 2 else grads())]
    tot = lambda a, b: float(torch.sqrt(sum(((x - y).double() ** 2).sum() for x, y in zip(a, b))) / torch.sqrt(sum((x.double() ** 2).sum() for x in a)))
    per = sorted(((float((x - y).norm() / (x.norm() + 1e-30)), n) for x, y, n in zip(e1, g, names)), reverse=True)[:4]
    print(f"step {i}: eager-eager {tot(e1, e2):.2e}  eager-graph {tot(e1, g):.2e}  consist {float(logs['warp_consist']):.5f}  worst params {[(round(a, 3), n) for a, n in per]}")
