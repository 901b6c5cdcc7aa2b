"""Cold / warm streaming bandwidth of this GPU for the footprints of the hot-path kernels (what a plain float4
kernel reaches: the practical floor under the 'cold' column of bench.py's kernel table).  Builds
scripts/probe/bw_probe.hip with hipcc into /tmp.   python scripts/bw_probe.py"""
import ctypes
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

here = os.path.dirname(os.path.abspath(__file__))
so = "/tmp/libbw_probe.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-o", so,
                       os.path.join(here, "probe", "bw_probe.hip")])
lib = ctypes.CDLL(so)
dev = torch.device("cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
flush = torch.zeros(768 * 1024 * 1024 // 4, dtype=torch.float32, device=dev)
out = torch.zeros(4, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
for mb in (34, 67, 134, 201, 268, 604):
    x = torch.randn(mb * 1000 * 1000 // 4, device=dev)
    nb = x.numel() * 4
    best = {}
    for blocks in (2048, 8192, 32768):
        for unroll in (1, 4, 8):
            fn = lambda: lib.probe_read(P(x), ctypes.c_int64(nb), P(out), blocks, unroll, st)
            c, w = bench.event_time_ms(fn, 15, flush=flush) * 1e3, bench.event_time_ms(fn, 15) * 1e3
            if not best or c < best["c"]:
                best = dict(c=c, w=w, cfg=(blocks, unroll))
    fnw = lambda: lib.probe_write(P(x), ctypes.c_int64(nb), 8192, st)
    cw, ww = bench.event_time_ms(fnw, 15, flush=flush) * 1e3, bench.event_time_ms(fnw, 15) * 1e3
    print(f"{mb:4d} MB  read cold {best['c']:6.1f} us ({nb / best['c'] / 1e6:5.2f} TB/s) warm {best['w']:6.1f} us ({nb / best['w'] / 1e6:5.2f} TB/s) "
          f"[blocks, unroll = {best['cfg']}]   write cold {cw:6.1f} us ({nb / cw / 1e6:5.2f} TB/s) warm {ww:6.1f} us ({nb / ww / 1e6:5.2f} TB/s)")
