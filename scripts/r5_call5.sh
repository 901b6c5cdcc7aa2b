#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/c5
mkdir -p $OUT
cd $ROOT
export HOC_KERNEL_GROUPS="render_backward_full(D+E+F);render_backward_train(E)"
bash scripts/prof_kernels.sh c5def $ROOT/bench.py --kernels-only > $OUT/def_by_grid.txt 2>&1
bash scripts/pmc_kernel.sh c5g1 "gather_kernel" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" $ROOT/bench.py --kernels-only > $OUT/pmc_gather_1.txt 2>&1
bash scripts/pmc_kernel.sh c5g2 "gather_kernel" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" $ROOT/bench.py --kernels-only > $OUT/pmc_gather_2.txt 2>&1
bash scripts/pmc_kernel.sh c5d1 "pixel_map_strip_kernel" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" $ROOT/bench.py --kernels-only > $OUT/pmc_strip_1.txt 2>&1
bash scripts/pmc_kernel.sh c5d2 "pixel_map_strip_kernel" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" $ROOT/bench.py --kernels-only > $OUT/pmc_strip_2.txt 2>&1
cat $OUT/def_by_grid.txt | grep -v "^$" | head -30
for f in $OUT/pmc_*.txt; do echo $f; tail -n 9 $f; done
