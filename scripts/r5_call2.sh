#!/bin/bash
# two failing tests in full + phase stamps
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/c2
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_graph_step.py "tests/test_gpu_bench.py::test_bench_single_process_line" -m gpu -q 2>&1 | tail -60 > $OUT/pytest.txt
bash scripts/r5_timeline.sh c2tl > $OUT/timeline_stdout.txt 2>&1
tail -5 $OUT/pytest.txt
