"""Kernel launches of the hot path, by kernel name (torch profiler)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from handobjectconsist_amd.models.synthnet import SynthMeshRegNet
from handobjectconsist_amd.models.warpreg import WarpRegNet
from handobjectconsist_amd.models import warpbranch
from handobjectconsist_amd.netscripts.epochpassconsist import SyntheticConsistLoader
dev = torch.device("cuda:0")
B, is_ = 64, 256
model = SynthMeshRegNet().to(dev).eval()
pre = WarpRegNet((is_, is_), model, lambda_consist=0.001, lambda_data=0.999, mano_faces=model.mano_layer.th_faces, pair_outputs="loss").to(dev)
loader = SyntheticConsistLoader(B, is_, seed=0, device=dev, pool=1)
_, consist = loader.step_batches(0)
def fake_results():
    out = []
    for s in consist["data"]:
        out.append({"recov_handverts3d": s["_handverts3d"].clone().requires_grad_(True), "recov_objverts3d": s["_objverts3d"].clone().requires_grad_(True)})
    return out
def hot():
    res = fake_results()
    l, _ = warpbranch.forward(consist["data"], res, pre.th_faces, pre.renderer, (is_, is_), pre.criterion, gt_refs=True,
                              hand_ignore_faces=pre.hand_ignore_faces, use_backward=True, pair_outputs="loss")
    l.backward()
for _ in range(3): hot()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=os.environ.get("HOC_ATEN_OPS", "0") == "1") as prof:
    hot()
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
cnt, tim = collections.Counter(), collections.Counter()
for e in evs:
    cnt[e.name[:70]] += 1; tim[e.name[:70]] += e.device_time if hasattr(e, "device_time") else e.cuda_time
print("launches", len(evs), "total device us", sum(tim.values()))
for k, v in sorted(tim.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{cnt[k]:4d} x {v:9.1f} us  {k}")

if os.environ.get("HOC_HOST_PROFILE", "0") == "1":
    # host side of the hot path: wall time per call with the GPU idle in between, and a cProfile of 30 calls
    import cProfile, pstats, time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        hot()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"30 calls: host issue time {t_issue / 30 * 1e6:.0f} us per call, with the GPU drained {t_all / 30 * 1e6:.0f} us per call")
    # host time inside each C-ABI call (launches are asynchronous: this is argument marshalling + hipLaunchKernel)
    from handobjectconsist_amd import _lib as L
    real, spent = L.call, collections.Counter()

    def timed(name, *a):
        t = time.perf_counter()
        try:
            return real(name, *a)
        finally:
            spent[name] += time.perf_counter() - t

    L.call = timed
    for _ in range(30):
        hot()
    torch.cuda.synchronize()
    L.call = real
    for k, v in spent.most_common():
        print(f"    {v / 30 * 1e6:7.1f} us per call inside _lib.call({k})")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(30):
        hot()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)

if os.environ.get("HOC_ATEN_OPS", "0") == "1":
    # which ATen operators of the pass launch something (the launches that are not this library's kernels: copies, fills, reductions)
    ops = collections.Counter()
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and len(getattr(e, "kernels", [])) > 0:
            ops[e.name + " " + str([tuple(s) if s else s for s in (e.input_shapes or [])][:3])] += 1
    for k, v in ops.most_common(30):
        print(f"{v:4d} x {k}")
    print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=60))
