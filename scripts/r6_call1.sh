#!/bin/bash
# round 6, first GPU call: the compact bench line (driver's command), GPU suite
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_call1
mkdir -p $OUT
cd $ROOT
python3 bench.py --gpus 1 --steps 20 --warmup 5 --details-out $OUT/bench_details.json > $OUT/bench_stdout.txt 2> $OUT/bench.err
tail -1 $OUT/bench_stdout.txt > $OUT/r06_bench_line.json
wc -c $OUT/r06_bench_line.json
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $OUT/pytest_gpu_tail.txt
tail -3 $OUT/pytest_gpu_tail.txt
cut -c1-600 $OUT/r06_bench_line.json
