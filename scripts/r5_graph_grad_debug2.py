"""Eager model vs graph-replayed twin, parameter by parameter.  python scripts/r5_graph_grad_debug2.py [is] [B] [cudnn_benchmark]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
from test_gpu_graph_step import _build
from handobjectconsist_amd.netscripts.epochpassconsist import GraphedTrainStep, train_step
is_ = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
torch.backends.cudnn.benchmark = (sys.argv[3] == "1") if len(sys.argv) > 3 else False
dev = torch.device("cuda:0")
m_e, pre_e, opt_e, ld_e = _build(dev, B, is_, 11, True, lr=0.0)
m_g, pre_g, opt_g, ld_g = _build(dev, B, is_, 11, True, lr=0.0)
names = [n for n, p in m_e.named_parameters() if p.requires_grad]
pe = [p for g in opt_e.param_groups for p in g["params"]]
step_g = GraphedTrainStep(pre_g, opt_g, experimental=True)
for i in range(8):
    le, _ = train_step(ld_e.step_batches(i), pre_e, opt_e)
    ge = [p.grad.detach().clone() for p in pe]
    try:
        lg, _ = step_g(ld_g.step_batches(i))
    except ValueError as e:
        print("step", i, "raised", str(e)[:40]); break
    gg = [x.detach().clone() for x in (step_g.last_grads if i >= 2 else [p.grad for g in opt_g.param_groups for p in g["params"]])]
    per = sorted(((float((x - y).norm() / (x.norm() + 1e-30)), n) for x, y, n in zip(ge, gg, names)), reverse=True)[:3]
    tot = float(torch.sqrt(sum(((x - y).double() ** 2).sum() for x, y in zip(ge, gg))) / torch.sqrt(sum((x.double() ** 2).sum() for x in ge)))
    print(f"step {i}: loss e {float(le):.6f} g {float(lg):.6f}  grads rel {tot:.2e}  worst {[(float('%.3g' % a), n) for a, n in per]}")
