"""The N > 1 path on CPU: two gloo ranks, batch-sharded data parallelism.

The render / warp kernels need no collective; the only exchange is the gradient all-reduce of
the encoder + heads (netscripts/gradreduce.BucketedGradReducer: buckets issued from inside
backward), with the reference's step structure (several forward passes -- data batch, then both
frames of the consist batch -- accumulated into ONE backward / optimiser step)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from handobjectconsist_amd.models.synthnet import SynthMeshRegNet
    from handobjectconsist_amd.netscripts.epochpassconsist import SyntheticConsistLoader
    from handobjectconsist_amd.netscripts.gradreduce import BucketedGradReducer

    torch.manual_seed(rank)  # replicas start DIFFERENT: the reducer's initial broadcast makes them rank 0's
    model = SynthMeshRegNet().eval()
    # BN statistics are frozen (--freeze_batchnorm): no buffer exchange.  Small buckets so that several are
    # issued from inside backward() and a few only by finish()
    reducer = BucketedGradReducer(model.parameters(), bucket_mb=2)
    assert len(reducer.buckets) > 4
    net = model
    opt = torch.optim.SGD(model.parameters(), lr=1e-3)
    ld = SyntheticConsistLoader(2, 64, seed=rank, device="cpu", pool=1)  # distinct shard per rank
    data, consist = ld.step_batches(0)
    # encoder + ONE pass of the heads / MANO over the three frames of the step (WarpRegNet.prepare); the three
    # per-frame forwards below only add the loss terms
    frames = [data["data"][0]] + list(consist["data"])
    for sample, chunk in zip(frames, net(frames, encode_only=True, batch_encoder=True)):
        sample["_post"] = chunk
    # epochpassconsist.py:57-68 structure: three forwards, one backward
    losses = [net(data["data"][0])[0]]
    for sample in consist["data"]:
        losses.append(0.5 * net(sample)[0])
    opt.zero_grad(set_to_none=True)
    (torch.stack([l.flatten() for l in losses]).sum() * reducer.loss_scale).backward()
    launched_in_backward = sum(b.launched for b in reducer.buckets)
    reducer.finish()
    assert launched_in_backward == len(reducer.buckets), "every bucket is issued by the hook of its last gradient"
    g = torch.cat([p.grad.flatten() for p in model.parameters()])
    opt.step()
    w = torch.cat([p.detach().flatten() for p in model.parameters()])
    gathered = [torch.zeros_like(w) for _ in range(world)]
    dist.all_gather(gathered, w)
    ggrads = [torch.zeros_like(g) for _ in range(world)]
    dist.all_gather(ggrads, g)
    if rank == 0:
        torch.save({"w": gathered, "g": ggrads}, out)
    dist.destroy_process_group()


def _local_grad(rank):
    from handobjectconsist_amd.models.synthnet import SynthMeshRegNet
    from handobjectconsist_amd.netscripts.epochpassconsist import SyntheticConsistLoader

    torch.manual_seed(0)
    model = SynthMeshRegNet().eval()
    ld = SyntheticConsistLoader(2, 64, seed=rank, device="cpu", pool=1)
    data, consist = ld.step_batches(0)
    losses = [model(data["data"][0])[0]] + [0.5 * model(s)[0] for s in consist["data"]]
    torch.stack([l.flatten() for l in losses]).sum().backward()
    return torch.cat([p.grad.flatten() for p in model.parameters()])


@pytest.mark.timeout(600)
def test_ddp_gloo_two_ranks(tmp_path):
    out = str(tmp_path / "res.pt")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out)
    # replicas stay in sync after the step
    assert torch.equal(res["w"][0], res["w"][1])
    assert torch.equal(res["g"][0], res["g"][1])
    # and the synchronised gradient is the mean of the per-shard gradients (computed with the
    # reference's one-encoder-pass-per-frame structure: the batched pass gives the same result)
    torch.set_num_threads(4)
    expect = (_local_grad(0) + _local_grad(1)) / 2
    err = (res["g"][0] - expect).abs().max() / expect.abs().max()
    assert err < 1e-4, err
