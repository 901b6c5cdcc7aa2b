"""Convolutions of the ResNet-18 trunk alone (forward + backward-data + backward-weight through autograd), NCHW vs
channels-last activations, at the step's batch (3 x 64 frames of 256 x 256): does MIOpen's NHWC path beat its NCHW
choice (Winograd + implicit GEMM with transposes) once BatchNorm is out of the picture?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
N = 192
# (C_in, C_out, k, stride, H_in, count in the trunk, needs grad wrt input)
LAYERS = [(3, 64, 7, 2, 256, 1, False), (64, 64, 3, 1, 64, 4, True), (64, 128, 3, 2, 64, 1, True), (128, 128, 3, 1, 32, 3, True),
          (64, 128, 1, 2, 64, 1, True), (128, 256, 3, 2, 32, 1, True), (256, 256, 3, 1, 16, 3, True), (128, 256, 1, 2, 32, 1, True),
          (256, 512, 3, 2, 16, 1, True), (512, 512, 3, 1, 8, 3, True), (256, 512, 1, 2, 16, 1, True)]
dtype = torch.bfloat16 if len(sys.argv) > 1 and sys.argv[1] == "bf16" else torch.float32
tot = {"nchw": 0.0, "nhwc": 0.0}
for cin, cout, k, s, h, cnt, need_dx in LAYERS:
    row = []
    for fmt in ("nchw", "nhwc"):
        mf = torch.channels_last if fmt == "nhwc" else torch.contiguous_format
        conv = torch.nn.Conv2d(cin, cout, k, s, k // 2, bias=False).to(dev, dtype).to(memory_format=mf)
        x = torch.randn(N, cin, h, h, device=dev, dtype=dtype).contiguous(memory_format=mf).requires_grad_(need_dx)
        y = conv(x)
        gy = torch.randn_like(y)

        def step():
            out = conv(x)
            out.backward(gy)
            conv.weight.grad = None
            if need_dx:
                x.grad = None

        for _ in range(3):
            step()
        ms = bench.event_time_ms(step, 10)
        row.append(ms)
        tot[fmt] += ms * cnt
    print(f"{cin:4d}->{cout:4d} k{k} s{s} {h:3d}px x{cnt}:  NCHW {row[0]:7.3f} ms   NHWC {row[1]:7.3f} ms")
print(f"trunk convolutions, fwd + bwd, per step: NCHW {tot['nchw']:.2f} ms, channels-last {tot['nhwc']:.2f} ms ({dtype})")
