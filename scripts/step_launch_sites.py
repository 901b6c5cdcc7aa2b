"""ATen operator calls of one training step's FORWARD side grouped by the line of this repository that issued them
(a TorchDispatchMode + the Python stack): where the short-launch tail of a step comes from (every small forward op
has one or more backward kernels behind it).  GPU box: python scripts/step_launch_sites.py"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

from handobjectconsist_amd.models.synthnet import SynthMeshRegNet
from handobjectconsist_amd.models.warpreg import WarpRegNet
from handobjectconsist_amd.netscripts.epochpassconsist import SyntheticConsistLoader, train_step

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
B, is_ = int(os.environ.get("B", 64)), 256
model = SynthMeshRegNet().to(dev).eval()
pre = WarpRegNet((is_, is_), model, lambda_consist=0.001, lambda_data=0.999, criterion="l1", gt_refs=True, progressive_steps=1000,
                 use_backward=True, mano_faces=model.mano_layer.th_faces, pair_outputs="loss").to(dev)
pre.step_count = 1000
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=5e-5, fused=True)
loader = SyntheticConsistLoader(B, is_, seed=0, device=dev, pool=2)
for i in range(4):
    train_step(loader.step_batches(i), pre, opt)
torch.cuda.synchronize()

VIEW_OPS = ("view", "reshape", "_unsafe_view", "expand", "slice", "select", "unsqueeze", "squeeze", "t.", "transpose", "permute",
            "detach", "alias", "as_strided", "unbind", "split", "_reshape_alias", "narrow", "unflatten", "flatten", "chunk")
sites = collections.defaultdict(collections.Counter)


class Sites(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__ if hasattr(func, "__name__") else str(func)
        full = str(func)
        if not any(full.split(".")[1].startswith(v.rstrip(".")) for v in VIEW_OPS):
            frame = "(no frame of this repository: autograd engine / optimiser)"
            for fs in reversed(traceback.extract_stack()[:-1]):
                if fs.filename.startswith(ROOT) and "/scripts/" not in fs.filename:
                    frame = f"{fs.filename[len(ROOT) + 1:]}:{fs.lineno}  {fs.line}"
                    break
            sites[frame][full.replace("aten.", "")] += 1
        return func(*args, **(kwargs or {}))


with Sites():
    train_step(loader.step_batches(0), pre, opt)
    torch.cuda.synchronize()
total = sum(sum(c.values()) for c in sites.values())
print(f"{total} non-view ATen calls in the step (forward side by line; backward / optimiser calls have no repository frame)")
for frame, ops in sorted(sites.items(), key=lambda kv: -sum(kv[1].values())):
    print(f"{sum(ops.values()):4d}  {frame[:150]}")
    print("        " + ", ".join(f"{c} x {o}" for o, c in ops.most_common(8)))
