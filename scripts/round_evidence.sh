#!/bin/bash
# Everything profiles/ holds for a round, in one GPU call:  scripts/round_evidence.sh r04   (GPU box; ~15 min)
# un-profiled bench line, kernel benches at the config-3 / config-5 raster sizes, rocprofv3 stats of the default bench
# command + one training step + PMC passes, the GPU test suite's tail.  Results under gpurun_out/evidence_<tag>/.
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/evidence_$TAG
mkdir -p $OUT
cd $ROOT
python3 bench.py --gpus 1 --steps 20 --warmup 5 --details-out $OUT/${TAG}_bench_details.json > $OUT/bench_stdout.txt 2> $OUT/bench.err
tail -1 $OUT/bench_stdout.txt > $OUT/${TAG}_bench_line.json   # (the driver's command; the contract line is the LAST stdout line)
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
bash scripts/hot_kernels.sh ${TAG}_metric > /dev/null 2>&1; cp $ROOT/gpurun_out/hot_kernels_${TAG}_metric.txt $OUT/${TAG}_hot_kernels_metric.txt
bash scripts/hot_kernels.sh ${TAG}_config3 --batch 8 --image-size 480 --image-height 270 > /dev/null 2>&1; cp $ROOT/gpurun_out/hot_kernels_${TAG}_config3.txt $OUT/${TAG}_hot_kernels_config3.txt
cat $ROOT/gpurun_out/hot_kernels_${TAG}_metric.json $ROOT/gpurun_out/hot_kernels_${TAG}_config3.json > $OUT/${TAG}_hot_path_ms.json
cp $P/pmc/summary.txt $OUT/${TAG}_pmc_summary.txt
cp $P/pmc/traffic.json $OUT/${TAG}_pmc_traffic.json
python -m pytest tests -m gpu -q 2>&1 | tail -12 > $OUT/${TAG}_pytest_gpu_tail.txt
for f in one_rank_reducer_vs_plain bench_config3 bench_config5; do cp $ROOT/gpurun_out/evidence/$f.json $OUT/${TAG}_$f.json 2>/dev/null; done
tail -3 $OUT/${TAG}_pytest_gpu_tail.txt; cut -c1-200 $OUT/${TAG}_bench_line.json
