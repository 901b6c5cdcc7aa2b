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


def _worker_guard(rank, world, port, out):
    """NaN on ONE rank, and buckets that complete out of order."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from handobjectconsist_amd.netscripts import epochpassconsist as E
    from handobjectconsist_amd.netscripts.gradreduce import BucketedGradReducer

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            # registration order outer, inner; the data flows inner -> outer, so the gradient of `outer` (LAST bucket:
            # buckets are filled in reverse registration order) is complete BEFORE the gradient of `inner` (first bucket)
            self.outer = torch.nn.Linear(300, 1)
            self.inner = torch.nn.Linear(4, 300)
            self.bad = False

        def forward(self, batch):
            loss = self.outer(torch.tanh(self.inner(batch["x"]))).mean().reshape(1)
            if self.bad:
                loss = loss * float("nan")
            return loss, {}, None, None

    res = {}
    for fused in (False, True):
        torch.manual_seed(7 + rank)
        net = Net()
        reducer = BucketedGradReducer(net.parameters(), bucket_mb=0.001)  # 1 KB: one parameter tensor per bucket
        assert len(reducer.buckets) == 4
        order = []
        launch = reducer._launch
        reducer._launch = lambda b: (order.append(reducer.buckets.index(b)), launch(b))[1]
        opt = torch.optim.Adam(net.parameters(), lr=0.1, fused=fused)
        batch = {"data": [{}], "x": torch.randn(5, 4)}
        E.train_step([batch], net, opt, reducer=reducer)
        assert order == [0, 1, 2, 3], order  # ready order is 2/3 first: held until 0 and 1 went out
        w1 = torch.cat([p.detach().flatten() for p in net.parameters()])
        net.bad = rank == 1  # only rank 1 diverges
        raised = []
        try:
            E.train_step([batch], net, opt, reducer=reducer)   # synchronous check: raises here, on BOTH ranks
            net.bad = False
            E.train_step([batch], net, opt, reducer=reducer)   # device-side guard: raises when the next step starts
        except ValueError as e:
            raised.append(str(e))
        w2 = torch.cat([p.detach().flatten() for p in net.parameters()])
        steps = [float(st["step"]) for st in opt.state.values()]
        # the healthy rank must have skipped the update too, and both must still be able to talk to each other
        probe = torch.ones(1)
        dist.all_reduce(probe)
        res[fused] = {"raised": raised, "same": bool(torch.equal(w1, w2)), "steps": steps, "probe": float(probe), "w": w2}
        reducer.remove()
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    if rank == 0:
        torch.save(gathered, out)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_nan_on_one_rank_stops_every_rank_in_the_same_step(tmp_path):
    """A NaN loss on one rank is NaN gradients on all of them after the all-reduce: the guard has to be global
    (the flag rides through the last gradient bucket).  Also: collectives go out in bucket order whatever order the
    gradients complete in."""
    out = str(tmp_path / "guard.pt")
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker_guard, args=(2, port, out), nprocs=2, join=True)
    per_rank = torch.load(out, weights_only=False)
    for fused in (False, True):
        for r in per_rank:
            assert len(r[fused]["raised"]) == 1 and "nan" in r[fused]["raised"][0]
            assert r[fused]["same"], "the diverged step touched the parameters of a rank"
            assert all(s == 1.0 for s in r[fused]["steps"])
            assert r[fused]["probe"] == 2.0
        assert torch.equal(per_rank[0][fused]["w"], per_rank[1][fused]["w"])


def _worker_many(rank, world, port, out):
    """World sizes the node will be asked for (4, 8): a tiny model in several buckets, a rank whose gradients complete in
    another order than everybody else's, a NaN on one rank in a later step."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from handobjectconsist_amd.netscripts import epochpassconsist as E
    from handobjectconsist_amd.netscripts.gradreduce import BucketedGradReducer

    late_rank, nan_rank = world - 1, min(5, world - 2)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(6, 40)
            self.b = torch.nn.Linear(40, 40)
            self.c = torch.nn.Linear(40, 1)
            self.bad = False

        def forward(self, batch):
            x = batch["x"]
            if rank == late_rank:
                # `c` (next to the loss: its gradients complete FIRST everywhere else) also enters at the very start of this
                # rank's graph, so here its gradients complete LAST -- with a zero weight, so that the value is unchanged
                x = x + 0.0 * self.c.weight[:, :6].sum() + 0.0 * self.c.bias.sum()
            loss = self.c(torch.tanh(self.b(torch.tanh(self.a(x))))).mean().reshape(1)
            if self.bad:
                loss = loss * float("nan")
            return loss, {}, None, None

    torch.manual_seed(100 + rank)  # replicas start different; the reducer's broadcast makes them rank 0's
    net = Net()
    reducer = BucketedGradReducer(net.parameters(), bucket_mb=0.0005)  # ~0.5 KB: c, then b in pieces, then a
    nb = len(reducer.buckets)
    assert nb >= 4
    order, ready = [], []
    launch = reducer._launch
    reducer._launch = lambda b: (order.append(reducer.buckets.index(b)), launch(b))[1]
    seen = [0] * nb

    def note(bi, n):  # (the order the buckets' gradients become complete in)
        def hook(_p):
            seen[bi] += 1
            if seen[bi] == n:
                ready.append(bi)
        return hook

    for bi, b in enumerate(reducer.buckets):
        for p in b.params:
            p.register_post_accumulate_grad_hook(note(bi, len(b.params)))
    opt = torch.optim.Adam(net.parameters(), lr=0.05, fused=False)
    torch.manual_seed(rank)
    batch = {"data": [{}], "x": torch.randn(4, 6)}
    shard_grads = None
    for step in range(3):
        order.clear(); ready.clear()
        seen[:] = [0] * nb
        E.train_step([batch], net, opt, reducer=reducer)
        assert order == list(range(nb)), (rank, order)
        if step == 0:
            g = torch.cat([p.grad.flatten() for p in net.parameters()]).clone()
            shard_grads = g
    ready_first = list(ready)
    w_ok = torch.cat([p.detach().flatten() for p in net.parameters()]).clone()
    net.bad = rank == nan_rank
    raised = []
    try:
        E.train_step([batch], net, opt, reducer=reducer)  # synchronous check: raises on EVERY rank in this step
    except ValueError as e:
        raised.append(str(e))
    w_after = torch.cat([p.detach().flatten() for p in net.parameters()])
    probe = torch.ones(1)
    dist.all_reduce(probe)
    res = {"rank": rank, "raised": raised, "same": bool(torch.equal(w_ok, w_after)), "w": w_after, "g": shard_grads,
           "probe": float(probe), "ready": ready_first, "buckets": nb}
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    if rank == 0:
        torch.save(gathered, out)
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [4, 8])
def test_four_and_eight_ranks(tmp_path, world):
    """The rank counts of the scaling runs (1 / 2 / 4 / 8): every rank issues its buckets in index order although
    one rank's gradients complete in another order, the averaged gradient and the replicas are identical everywhere after
    three steps, and a NaN on ONE rank (rank 5 of 8) stops all of them in the same step with the group still usable."""
    out = str(tmp_path / f"many{world}.pt")
    port = 33500 + (os.getpid() % 2000) + world
    mp.spawn(_worker_many, args=(world, port, out), nprocs=world, join=True)
    per_rank = torch.load(out, weights_only=False)
    assert sorted(r["rank"] for r in per_rank) == list(range(world))
    for r in per_rank:
        assert torch.equal(r["w"], per_rank[0]["w"]), f"replica {r['rank']} diverged"
        assert torch.equal(r["g"], per_rank[0]["g"]), f"rank {r['rank']} holds another averaged gradient"
        assert len(r["raised"]) == 1 and "nan" in r["raised"][0].lower()
        assert r["same"], "the diverged step touched the parameters of a rank"
        assert r["probe"] == float(world)
    # the hold was exercised: the late rank's first bucket completes last, everybody else's first
    late = per_rank[world - 1]["ready"]
    assert late.index(0) > late.index(per_rank[0]["buckets"] - 1), late
    assert per_rank[0]["ready"].index(0) < per_rank[0]["ready"].index(per_rank[0]["buckets"] - 1)
