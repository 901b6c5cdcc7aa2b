"""bench.py's contract on one MI355X, launched the way the driver launches it: as a plain script (N=1) and
under ``torch.distributed.run`` through the RCCL process group + DistributedDataParallel code path (one rank
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


def _json_line(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_single_process_line(cuda):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + SMALL, capture_output=True,
                         text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    line = _json_line(res.stdout)
    for k in CONTRACT:
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1 and line["value"] > 0
    assert line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert "workload" in line["config"] and line["data"] == "synthetic" and line["dtype"] == "f32"
    roof, cpu = line["roofline"], line["cpu_baseline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert abs(roof["achieved"] - roof["algorithmic_bytes"] / (roof["launch_ms"] * 1e-3) / 1e9) <= 0.02 * roof["achieved"]
    assert cpu["kind"] == "port" and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["sample"]
    # the informational run of the same step with the stock trunk modules
    assert line["stock_trunk"]["value"] > 0 and line["stock_trunk"]["ms_per_step"] > 0


@pytest.mark.gpu
def test_bench_under_torchrun_rccl_ddp(cuda):
    env = dict(os.environ, HOC_FORCE_DDP="1", HOC_CHECK_REPLICAS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no-cpu-baseline",
           "--no-kernel-bench"] + SMALL
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    line = _json_line(res.stdout)
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["parallelism"] == "dp1"


@pytest.mark.gpu
def test_bench_two_ranks_share_the_gpu_over_gloo(cuda):
    """world_size 2 on the one-GPU box: both ranks on cuda:0, gradient all-reduce of DistributedDataParallel over gloo
    (RCCL refuses two ranks per device).  Exercises what N > 1 adds -- per-rank seeds and loaders, DDP's bucketed
    all-reduce through this build's autograd functions, the barrier + max-over-ranks timing, rank-0-only output;
    HOC_CHECK_REPLICAS makes bench.py assert that the replicas' parameters are bit-identical after the steps."""
    env = dict(os.environ, HOC_SHARE_GPU="1", HOC_DIST_BACKEND="gloo", HOC_CHECK_REPLICAS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29543", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline",
           "--no-kernel-bench"] + SMALL
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    line = _json_line(res.stdout)
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["config"]["parallelism"] == "dp2"
    assert line["config"]["global_batch"] == 8 and line["cpu_baseline"] is None
