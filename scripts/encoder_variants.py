"""Experiment: stock-PyTorch ResNet-18 fwd+bwd time at the bench shapes under different settings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from handobjectconsist_amd.models.synthnet import ResNet18Features

dev = torch.device("cuda:0")
def run(name, B, calls, channels_last=False, benchmark=False, dtype=None, iters=5):
    torch.backends.cudnn.benchmark = benchmark
    m = ResNet18Features().to(dev).eval()
    x = torch.randn(B, 3, 256, 256, device=dev)
    if channels_last:
        m = m.to(memory_format=torch.channels_last); x = x.contiguous(memory_format=torch.channels_last)
    def step():
        loss = 0
        for _ in range(calls):
            with torch.autocast("cuda", dtype=dtype, enabled=dtype is not None):
                loss = loss + m(x).float().sum()
        loss.backward()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): step()
    torch.cuda.synchronize()
    print(f"{name:50s} {(time.perf_counter() - t0) / iters * 1e3:8.2f} ms", flush=True)

run("fp32 nchw, 3 calls of 64", 64, 3)
run("fp32 nchw, 1 call of 192", 192, 1)
run("fp32 nchw benchmark=True, 3x64", 64, 3, benchmark=True)
run("fp32 channels_last, 3x64", 64, 3, channels_last=True)
run("fp32 channels_last benchmark, 3x64", 64, 3, channels_last=True, benchmark=True)
run("fp32 channels_last benchmark, 1x192", 192, 1, channels_last=True, benchmark=True)
run("bf16 autocast channels_last, 3x64", 64, 3, channels_last=True, dtype=torch.bfloat16)
run("bf16 autocast nchw, 3x64", 64, 3, dtype=torch.bfloat16)
