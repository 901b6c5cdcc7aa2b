"""ResNet-18 trunk fwd+bwd, 3 passes of B=64 at 256x256: NCHW vs channels_last (GPU ms by events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from handobjectconsist_amd.models.synthnet import SynthMeshRegNet
dev = torch.device("cuda:0")
for name, cl, bench_flag in (("NCHW", False, False), ("channels_last", True, False), ("NCHW + cudnn.benchmark", False, True), ("channels_last + cudnn.benchmark", True, True)):
    torch.backends.cudnn.benchmark = bench_flag
    torch.manual_seed(0)
    net = SynthMeshRegNet().to(dev).eval().base_net
    x = [torch.rand(64, 3, 256, 256, device=dev) - 0.5 for _ in range(3)]
    if cl:
        net = net.to(memory_format=torch.channels_last)
        x = [xi.contiguous(memory_format=torch.channels_last) for xi in x]
    def three():
        loss = sum(net(xi).sum() for xi in x)
        loss.backward()
    for _ in range(4): three()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): three()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:34s} gpu {e0.elapsed_time(e1) / 5:7.2f} ms per step-equivalent")
