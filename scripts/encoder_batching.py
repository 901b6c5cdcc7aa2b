"""ResNet-18 trunk fwd+bwd: 3 passes of B=64 vs one pass of B=192 (GPU time by events, CPU enqueue time)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from handobjectconsist_amd.models.synthnet import SynthMeshRegNet
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = SynthMeshRegNet().to(dev).eval().base_net
x = [torch.rand(64, 3, 256, 256, device=dev) - 0.5 for _ in range(3)]
xc = torch.cat(x)
def three():
    loss = sum(net(xi).sum() for xi in x)
    loss.backward()
def one():
    net(xc).sum().backward()
x128 = torch.cat(x[1:])
def two():
    (net(x[0]).sum() + net(x128).sum()).backward()
for name, fn in (("3 x B=64", three), ("64 + 128", two), ("1 x B=192", one), ("3 x B=64", three), ("64 + 128", two), ("1 x B=192", one)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(5): fn()
    e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize()
    print(f"{name:10s} gpu {e0.elapsed_time(e1) / 5:7.2f} ms   cpu enqueue {(t1 - t0) / 5 * 1e3:7.2f} ms")
