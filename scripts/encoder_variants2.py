import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from handobjectconsist_amd.models.synthnet import ResNet18Features
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = os.environ.get("BENCHMARK", "0") == "1"
cl = os.environ.get("CL", "0") == "1"
m = ResNet18Features().to(dev).eval(); x = torch.randn(64, 3, 256, 256, device=dev)
if cl:
    m = m.to(memory_format=torch.channels_last); x = x.contiguous(memory_format=torch.channels_last)
def step():
    loss = 0
    for _ in range(3): loss = loss + m(x).sum()
    loss.backward()
for _ in range(4): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize()
print(f"FIND_MODE={os.environ.get('MIOPEN_FIND_MODE')} BENCHMARK={torch.backends.cudnn.benchmark} CL={cl}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms", flush=True)
