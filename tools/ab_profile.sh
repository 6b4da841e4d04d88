#!/bin/bash
# A/B of the two j-operand paths at 65 536 bodies under rocprofv3 (run on the GPU box).
R=$PWD; mkdir -p gpurun_out/ab; cd /tmp; export TMPDIR=/tmp
for plan in sgpr,2,8,8,0 lds,2,8,8,0 lds,4,8,8,0 sgpr,4,8,8,0; do
  tag=${plan//,/_}
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab/stats_$tag -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --plan $plan > $R/gpurun_out/ab/bench_$tag.txt 2>&1
done
for plan in sgpr,2,8,8,0 lds,2,8,8,0; do
  tag=${plan//,/_}
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/gpurun_out/ab/pmc_$tag -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --plan $plan > /dev/null 2>&1
done
cd $R
for d in gpurun_out/ab/stats_*; do echo "== $d"; cat $(find $d -name "*kernel_stats.csv" | head -1) | head -3; done
