#!/bin/bash
# round 6, call 9: full GPU suite + the driver's bench command on the build with the pair step / item table / DPP scans
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_call9
mkdir -p $OUT
cd $ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $OUT/pytest_gpu_tail.txt
tail -4 $OUT/pytest_gpu_tail.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 --details-out $OUT/bench_details.json > $OUT/bench_stdout.txt 2> $OUT/bench.err
tail -1 $OUT/bench_stdout.txt > $OUT/r06_bench_line.json
wc -c $OUT/r06_bench_line.json; cat $OUT/r06_bench_line.json
tail -5 $OUT/bench.err
