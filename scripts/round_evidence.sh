#!/bin/bash
# Everything profiles/ holds for a round, in one GPU call:  scripts/round_evidence.sh r04   (GPU box; ~15 min)
# un-profiled bench line, kernel benches at the config-3 / config-5 raster sizes, rocprofv3 stats of the default bench
# command + one training step + PMC passes, the GPU test suite's tail.  Results under gpurun_out/evidence_<tag>/.
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/evidence_$TAG
mkdir -p $OUT
cd $ROOT
python bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/bench.err
python bench.py --kernels-only --batch 8 --image-size 480 > $OUT/${TAG}_kernels_480.json 2>/dev/null
python bench.py --kernels-only --batch 32 --image-size 640 > $OUT/${TAG}_kernels_640.json 2>/dev/null
bash scripts/prof_round.sh $TAG > /dev/null 2>&1
bash scripts/prof_step.sh $TAG > /dev/null 2>&1
P=$ROOT/gpurun_out/prof_$TAG
cp $P/bench_kernel_stats.csv $OUT/${TAG}_bench_kernel_stats.csv
cp $P/bench_mr_kernels.txt $OUT/${TAG}_bench_mr_kernels.txt
cp $P/bench_mr_kernels_by_grid.txt $OUT/${TAG}_bench_mr_kernels_by_grid.txt
cp $P/step_kernels.txt $OUT/${TAG}_step_kernels.txt
cp $P/step_sequence.txt $OUT/${TAG}_step_sequence.txt
cp $P/hot_path_launches.txt $OUT/${TAG}_hot_path_launches.txt
cp $P/pmc/summary.txt $OUT/${TAG}_pmc_summary.txt
cp $P/pmc/traffic.json $OUT/${TAG}_pmc_traffic.json
python -m pytest tests -m gpu -q 2>&1 | tail -12 > $OUT/${TAG}_pytest_gpu_tail.txt
for f in one_rank_reducer_vs_plain bench_config3 bench_config5; do cp $ROOT/gpurun_out/evidence/$f.json $OUT/${TAG}_$f.json 2>/dev/null; done
tail -3 $OUT/${TAG}_pytest_gpu_tail.txt; cut -c1-200 $OUT/${TAG}_bench_line.json
