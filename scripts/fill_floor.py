"""How fast can this GPU write / read N MB?  (floor for the background fill of the rasteriser)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda:0")
flush = torch.zeros(768 * 1024 * 1024 // 4, dtype=torch.float32, device=dev)
for mb in (17, 50, 151, 604):
    x = torch.empty(mb * 1024 * 1024 // 4, dtype=torch.float32, device=dev)
    y = torch.empty_like(x)
    for name, fn, nbytes in (("fill", lambda: x.fill_(1.0), x.numel() * 4), ("copy", lambda: y.copy_(x), x.numel() * 8),
                             ("sum", lambda: x.sum(), x.numel() * 4)):
        c = bench.event_time_ms(fn, 20, flush=flush) * 1e3
        w = bench.event_time_ms(fn, 20) * 1e3
        print(f"{mb:4d} MB {name:5s} cold {c:7.1f} us ({nbytes / c / 1e6:6.2f} TB/s)   warm {w:7.1f} us ({nbytes / w / 1e6:6.2f} TB/s)")
