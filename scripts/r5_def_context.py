"""D + E + F of the kernel bench after different process histories: nothing / eager training steps / graph-replayed steps.
    python scripts/r5_def_context.py none|eager|graph"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
mode = sys.argv[1]
dev = torch.device("cuda:0")
if mode != "none":
    from handobjectconsist_amd.models.synthnet import SynthMeshRegNet
    from handobjectconsist_amd.models.warpreg import WarpRegNet
    from handobjectconsist_amd.netscripts import epochpassconsist as E
    torch.backends.cudnn.benchmark = True
    model = SynthMeshRegNet().to(dev).eval().to(memory_format=torch.channels_last)
    pre = WarpRegNet((256, 256), model, lambda_consist=0.001, lambda_data=0.999, criterion="l1", gt_refs=True, progressive_steps=1000,
                     use_backward=True, mano_faces=model.mano_layer.th_faces, pair_outputs="loss").to(dev)
    pre.step_count = 1000
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=5e-5, fused=True, capturable=True)
    loader = E.SyntheticConsistLoader(64, 256, seed=0, device=dev, pool=2)
    step = E.GraphedTrainStep(pre, opt) if mode == "graph" else (lambda b: E.train_step(b, pre, opt))
    for i in range(20):
        step(loader.step_batches(i))
    torch.cuda.synchronize()
    if len(sys.argv) > 2 and sys.argv[2] == "free":
        del step, loader, opt, pre, model
        import gc; gc.collect(); torch.cuda.empty_cache()
out = bench.kernel_bench(dev, 64, 256, 10, ("render_backward_full(D+E+F)", "render_backward_train(E)"))
print(mode, sys.argv[2:], {k[:28]: (v["ms"], v["ms_cache_warm"]) for k, v in out.items()}, "reserved MB", torch.cuda.memory_reserved() // 2**20)
