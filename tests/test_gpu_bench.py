"""bench.py's contract on one MI355X, launched the way the driver launches it: as a plain script (N=1) and
under ``torch.distributed.run`` through the RCCL process group + bucketed gradient reducer code path (one rank
here -- the box has a single GPU; two gloo ranks on CPU are in test_dist_gloo.py)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--batch", "4", "--image-size", "64", "--steps", "2", "--warmup", "1", "--kernel-iters", "2", "--cpu-sample", "2"]
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config")


EVIDENCE = os.path.join(ROOT, "gpurun_out", "evidence")  # scratch; merged back from the GPU box, copied to profiles/


def _keep(name, line):
    os.makedirs(EVIDENCE, exist_ok=True)
    with open(os.path.join(EVIDENCE, name), "w") as fh:
        fh.write(json.dumps(line) + "\n")


def _json_line(out):
    """stdout of bench.py: ONE line, the compact contract line -- and it is the last line, at most 4 KB (what the driver's
    record keeps and parses; round 5's 21 KB line was lost there)."""
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    assert out.strip().splitlines()[-1] == lines[0] and len(lines[0].encode()) <= 4096, len(lines[0])
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_single_process_line(cuda, tmp_path):
    details = str(tmp_path / "details.json")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--details-out", details] + SMALL,
                         capture_output=True, text=True, timeout=1800, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    line = _json_line(res.stdout)
    for k in CONTRACT:
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1 and line["value"] > 0
    assert line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert "workload" in line["config"] and line["data"] == "synthetic" and line["dtype"] == "f32"
    assert len(line["unit"]) <= 200 and line["details"] == details
    # --- the compact line: the figures the driver's record has to hold
    roof, cpu = line["roofline"], line["cpu_baseline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert abs(roof["achieved"] - roof["bytes"] / (roof["launch_ms"] * 1e-3) / 1e9) <= 0.02 * roof["achieved"]
    assert 0 < roof["frac"] < 1 and roof["bytes"] < roof["algorithmic_bytes"] and roof["frac_algorithmic"] > roof["frac"]
    assert 0 < roof["frac_cache_warm"] and roof["in_step_us"] > 0
    assert abs(roof["frac_in_step"] - roof["bytes"] / (roof["in_step_us"] * 1e-6) / 1e9 / roof["peak"]) < 1e-3
    assert roof["traffic"] is not None and roof["traffic"] >= roof["traffic_low"] > 0, "in-run PMC passes failed on the GPU box"
    assert roof["forward"]["frac"] > 0 and roof["forward"]["launch_ms"] > 0 and roof["forward"]["in_step_us"] > 0
    assert roof["d_e_f"]["ms"] > 0 and 0 < roof["d_e_f"]["frac"] < 1
    assert 0 < roof["d_e_f"]["frac_of_algorithmic_issue"] < 1 and roof["d_e_f"]["terms"] > 0
    assert roof["hot_path_device_ms"] > 0 and roof["hot_path_eager_ms_host_bound"] > 0 and roof["stock_trunk_it_s"] > 0
    assert cpu["kind"] == "port" and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["sample"] and len(cpu["sample"]) <= 200
    # --- the details file: everything else
    with open(details) as fh:
        full = json.load(fh)
    assert full["value"] == line["value"] and full["ms_per_step"] == line["ms_per_step"]
    froof = full["roofline"]
    # the raster backward as the step launches it: the scatter of the unit gradient the step's forward launch left
    assert froof["device_kernels"] == ["unit_scatter_tiles_kernel"] and froof["frac"] == roof["frac"]
    assert froof["recomputing_form"]["device_kernels"] == ["pair_scatter_tiles_kernel"] and 0 < froof["recomputing_form"]["frac"] < 1
    assert froof["scatter_alone"]["device_kernels"] == ["scatter_tiles_kernel<true, true>"] and 0 < froof["scatter_alone"]["frac"] < 1
    for name, w in full["warp_tiles"].items():
        assert 0 < w["frac"] < 1 and w["launch_ms"] > 0, name
    assert full["warp_tiles"]["flow_pair_forward_grad_tiles(train: occlusion + epilogue + pair loss + unit gradient, sparse)"]["in_step_us"] > 0
    fwd = full["roofline_forward"]
    assert fwd["bound"] == "hbm" and "raster_tile_kernel<true, true>" in fwd["device_kernels"] and fwd["launch_ms"] > 0
    assert fwd["bytes"] == fwd["algorithmic_bytes"]
    for r in (froof, fwd):  # HBM bytes from the PMC passes this very run made (None only if rocprofv3 is unavailable)
        if r["traffic"] is not None:
            assert r["traffic"] >= r["traffic_low"] > 0 and r["dram_frac"] > 0
            assert abs(r["dram_frac"] - r["traffic"] / (r["launch_ms"] * 1e-3) / 1e9 / r["peak"]) < 1e-3
    assert full["cpu_baseline"]["processes"] >= 1
    # the run of the same step with the stock trunk modules
    assert full["stock_trunk"]["value"] > 0 and full["stock_trunk"]["ms_per_step"] > 0


@pytest.mark.gpu
def test_bench_under_torchrun_rccl_ddp(cuda):
    env = dict(os.environ, HOC_FORCE_DDP="1", HOC_CHECK_REPLICAS="1", HOC_TUNABLEOP="0")
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)  # bench.py sets what RCCL needs itself
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no-cpu-baseline",
           "--no-kernel-bench"] + SMALL
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=1800, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    line = _json_line(res.stdout)
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["parallelism"] == "dp1"
    assert line["ranks"]["backend"] == "rccl" and "BucketedGradReducer" in line["ranks"]["reducer"]


@pytest.mark.gpu
def test_bench_two_ranks_share_the_gpu_over_gloo(cuda):
    """world_size 2 on the one-GPU box: both ranks on cuda:0, the bucketed gradient all-reduce over gloo
    (RCCL refuses two ranks per device).  Exercises what N > 1 adds -- per-rank seeds and loaders, the reducer's
    all-reduces issued from inside the backward pass of this build's autograd functions, the barrier + max-over-ranks timing, rank-0-only output;
    HOC_CHECK_REPLICAS makes bench.py assert that the replicas' parameters are bit-identical after the steps."""
    env = dict(os.environ, HOC_SHARE_GPU="1", HOC_DIST_BACKEND="gloo", HOC_CHECK_REPLICAS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", HOC_TUNABLEOP="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29543", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline",
           "--no-kernel-bench"] + SMALL
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=1800, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    line = _json_line(res.stdout)
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["config"]["parallelism"] == "dp2"
    assert line["config"]["global_batch"] == 8 and line["cpu_baseline"] is None
    assert line["ranks"]["world_size"] == 2 and len(line["ranks"]["ms_per_step_by_rank"]) == 2


@pytest.mark.gpu
def test_bench_gpus_2_as_a_plain_script_starts_its_own_ranks(cuda):
    """The driver's command shape -- ``python bench.py --gpus N ...``, a plain script, no launcher, no RANK / WORLD_SIZE
    in the environment: bench.py re-executes itself under torch.distributed.run with N ranks (here the two ranks share
    the box's one GPU over gloo) and rank 0's ONE JSON line is the command's stdout; replicas checked bit-identical."""
    env = dict(os.environ, HOC_SHARE_GPU="1", HOC_DIST_BACKEND="gloo", HOC_CHECK_REPLICAS="1", HOC_TUNABLEOP="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline", "--no-kernel-bench"] + SMALL
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=1800, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    line = _json_line(res.stdout)
    for k in CONTRACT:
        assert k in line, k
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["config"]["parallelism"] == "dp2"
    assert line["ranks"]["world_size"] == 2 and len(line["ranks"]["ms_per_step_by_rank"]) == 2
    assert "torch.distributed.run" in res.stderr  # the launcher line bench.py prints
    # without device sharing the same command must refuse with a message, not an assertion, on a one-GPU box
    import torch
    if torch.cuda.device_count() < 2:
        env.pop("HOC_SHARE_GPU")
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
        assert res.returncode != 0 and "--gpus 2 asks for 2 devices" in res.stderr and "Traceback" not in res.stderr


@pytest.mark.gpu
def test_bench_gpus_8_as_a_plain_script(cuda):
    """The command the scaling run will one day issue -- ``python bench.py --gpus 8`` -- executed once before a node with 8
    devices exists: 8 ranks started by bench.py itself, sharing this box's GPU(s) over gloo; 8 seeds / loaders, the
    in-order buckets on 8 ranks, the `ranks` block with 8 rows, the global batch of 8 shards, replicas bit-identical
    (HOC_CHECK_REPLICAS), rank 0's reduced kernel bench while the others wait.  Wall time bounded."""
    import time

    env = dict(os.environ, HOC_SHARE_GPU="1", HOC_DIST_BACKEND="gloo", HOC_CHECK_REPLICAS="1", HOC_TUNABLEOP="0", HOC_CUDNN_BENCHMARK="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--batch", "2", "--image-size", "64", "--steps", "2",
           "--warmup", "1", "--no-kernel-bench", "--no-cpu-baseline"]
    t0 = time.time()
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=1800, cwd=ROOT, env=env)
    wall = time.time() - t0
    assert res.returncode == 0, res.stderr[-3000:]
    line = _json_line(res.stdout)
    for k in CONTRACT:
        assert k in line, k
    assert line["n_gpus"] == 8 and line["value"] > 0 and line["config"]["parallelism"] == "dp8"
    assert line["config"]["global_batch"] == 16 and line["scaling"] == "weak"
    assert line["ranks"]["world_size"] == 8 and len(line["ranks"]["ms_per_step_by_rank"]) == 8
    assert all(ms > 0 for ms in line["ranks"]["ms_per_step_by_rank"])
    assert line["cpu_baseline"] is None and line["roofline"] is None  # (N = 1 legs only; --no-kernel-bench)
    assert wall < 300, f"bench.py --gpus 8 took {wall:.0f} s"


@pytest.mark.gpu
def test_one_rank_rccl_step_costs_what_the_plain_step_costs(cuda):
    """The data-parallel code path must not tax the step before a byte is communicated (round 2 measured torch's
    DistributedDataParallel wrapper at +13-15 % on one rank).  Headline workload (B = 64, 256 x 256).  Both loops run
    in ONE process (``bench.py --reducer-ab``): one model, one set of MIOpen solver / TunableOp choices -- round 3
    compared two processes, whose separate solver searches alone moved a pair by up to 5 % -- so every pair counts: the
    reducer's real cost on one rank is 1.8-2.5 % (its bucket copies and four -- nine until round 5 -- one-rank all-reduce launches; pairs of round 4:
    1.025, 1.026, 1.025, 1.032), so with +-0.5 % of block-to-block noise a bound of 1.03 on EACH of two pairs fails one run in
    a few.  Three pairs: the MEDIAN ratio <= 1.03 and no pair above 1.045 (not the minimum of the pairs, which round 3 used)."""
    bench = os.path.join(ROOT, "bench.py")
    res = subprocess.run([sys.executable, bench, "--gpus", "1", "--steps", "20", "--warmup", "6", "--reducer-ab", "3"],
                         capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    line = _json_line(res.stdout)
    _keep("one_rank_reducer_vs_plain.json", line)
    assert line["backend"] == "rccl" and line["buckets"] >= 3 and len(line["ratios"]) == 3
    ratios = sorted(red / plain for plain, red in zip(line["plain_ms"], line["one_rank_rccl_ms"]))
    assert ratios[1] <= 1.03 and ratios[2] <= 1.045, line


@pytest.mark.gpu
def test_config4_per_gpu_workload_under_rccl_ddp(cuda):
    """BASELINE config 4 (8 x MI355X, global B = 256): the per-GPU share -- B = 32 frame pairs of 256 x 256 -- through
    the RCCL process group + gradient reducer path (one rank: the test box has one GPU); the JSON line
    carries the per-rank evidence (backend, device, per-rank step time, all-reduce volume)."""
    env = dict(os.environ, HOC_FORCE_DDP="1", HOC_TUNABLEOP="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--batch", "32",
           "--image-size", "256", "--steps", "2", "--warmup", "2", "--no-cpu-baseline", "--no-kernel-bench",
           "--no-stock-trunk"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=1800, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    line = _json_line(res.stdout)
    assert line["config"]["global_batch"] == 32 and line["value"] > 0 and line["ms_per_step"] > 0
    ranks = line["ranks"]
    assert ranks["backend"] == "rccl" and ranks["world_size"] == 1 and len(ranks["ms_per_step_by_rank"]) == 1
    assert ranks["ms_per_step_by_rank"][0] > 0 and ranks["device_by_rank"] == [0]
    assert 47.0 < ranks["grad_allreduce_MB"] < 49.0  # 11.98 M trainable fp32 parameters


@pytest.mark.gpu
def test_config2_data_only_step(cuda):
    """BASELINE config 2 (trainmeshreg.py, B = 32): a fully supervised step -- encoder, heads, MANO layer, 2-D / 3-D
    losses, backward, Adam -- with no render / warp in it; runs through this build's MANO / head
    post-processing / trunk glue kernels."""
    import torch

    from handobjectconsist_amd.models.synthnet import SynthMeshRegNet
    from handobjectconsist_amd.models.warpreg import WarpRegNet
    from handobjectconsist_amd.netscripts import epochpassconsist as E

    torch.manual_seed(0)
    model = SynthMeshRegNet().to(cuda).eval()
    pre = WarpRegNet((256, 256), model, lambda_consist=0.001, lambda_data=0.999, criterion="l1", gt_refs=True,
                     use_backward=True, mano_faces=model.mano_layer.th_faces, pair_outputs="loss").to(cuda)
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=5e-5)
    loader = E.SyntheticConsistLoader(32, 256, seed=0, device=cuda, pool=1)
    data_batch = loader.step_batches(0)[0]
    assert data_batch["supervision"] == "data"
    before = [p.detach().clone() for p in model.parameters() if p.requires_grad][:4]
    losses = [float(E.train_step([data_batch], pre, opt)[0]) for _ in range(3)]
    E.raise_pending_nan(opt)  # (train_step's contract for direct callers: the last step's device-side flag)
    assert all(l == l and l > 0 for l in losses)
    assert losses[-1] < losses[0], losses  # three Adam steps on one batch reduce its loss
    after = [p.detach() for p in model.parameters() if p.requires_grad][:4]
    assert any(not torch.equal(a, b_) for a, b_ in zip(after, before))


@pytest.mark.gpu
@pytest.mark.parametrize("name,extra,dtype", [
    ("config3", ["--batch", "8", "--image-size", "480", "--image-height", "270"], "f32"),
    ("config5", ["--batch", "32", "--image-size", "640", "--image-height", "480", "--encoder-dtype", "bf16"], "bf16"),
])
def test_baseline_configs_3_and_5_full_steps(cuda, name, extra, dtype):
    """Full optimiser steps of BASELINE.json's config 3 (trainmeshwarp.py on 480 x 270 frame pairs: 480-pixel raster,
    cropped) and config 5 (640 x 480 frames, trunk under bf16 autocast, render / warp / heads in fp32) through bench.py:
    the line names the workload and the precision, the loss is finite (bench.py asserts it), and the line is kept as
    evidence (gpurun_out/evidence -> profiles/)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "2", "--no-kernel-bench",
           "--no-cpu-baseline", "--no-stock-trunk"] + extra
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    line = _json_line(res.stdout)
    w, h = extra[3], extra[5]
    assert f"{w}x{h}" in line["config"]["workload"] and f"B={extra[1]}" in line["config"]["workload"]
    assert line["config"]["image_size"] == int(w) and line["value"] > 0 and line["ms_per_step"] > 0
    assert line["dtype"].startswith(dtype) and ("bf16-autocast" in line["config"]["workload"]) == (dtype == "bf16")
    assert line["hot_path_ms"] > 0
    _keep(f"bench_{name}.json", line)


@pytest.mark.gpu
@pytest.mark.parametrize("batch,size,floor", [(32, 640, 0.60), (8, 480, 0.275)])
def test_forward_roofline_at_the_raster_sizes_of_configs_3_and_5(cuda, batch, size, floor):
    """The flow-mode forward of the training step at the raster sizes of BASELINE configs 5 (640 x 640, B = 32) and 3
    (480 x 480, B = 8): fraction of the 8 TB/s roofline on SURVEY 8(d)'s algorithmic bytes, cold caches.  640: >= 0.60 (0.65-
    0.67 measured in rounds 4 and 5, 0.70 in round 6; 0.31 before round 4, when one face spanning more than eight bins made every tile of its
    image a listed tile).  480 at B = 8 is 16 renders -- a launch too small to fill 256 compute units: 0.30-0.31 measured (60.7
    us; the binning pass in 16 parts per image since round 5 took 1.6 us off it: its per-face pass and the tile kernel's
    chain of phases per tile are what the launch consists of; 0.307 = 60.1 us in round 6 with the item table + DPP scans of
    the tile kernel), floor 0.275: ten per cent under the measurement (box to box the tile kernel differs by that much)."""
    env = dict(os.environ, HOC_KERNEL_GROUPS="render_flow_forward(train outputs,both frames=2B)")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--kernels-only", "--batch", str(batch), "--image-size",
                          str(size), "--kernel-iters", "20"], capture_output=True, text=True, timeout=1800, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    k = json.loads(res.stdout)["render_flow_forward(train outputs,both frames=2B)"]
    _keep(f"forward_roofline_{size}.json", k)
    assert k["frac_hbm_peak"] >= floor, k
