#!/bin/bash
# round 2, third GPU pass: whole fast suite (no -x), microbenchmarks incl. the large f32 MFMA shapes,
# default bench with the in-kernel clock, kernel trace + the four PMC passes of the ticket kernel.
R=$PWD; O=$R/gpurun_out/r02c; mkdir -p $O
python -m pytest tests -m "gpu and not slow" -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -25 $O/pytest_gpu.txt
tools/ubench 20000 > $O/ubench.txt 2>&1; tail -34 $O/ubench.txt
python bench.py --steps 200 --warmup 20 > $O/bench_default.json 2> $O/bench_default.err; cat $O/bench_default.json | cut -c1-1800
python bench.py --steps 20 --warmup 5 > $O/bench_driver_like.json 2> $O/bench_driver_like.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ubench_stats -- $R/tools/ubench 5000 > $O/ubench_profiled.txt 2>&1
mkdir -p $O/pmc
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc/fetch -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc/write -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/pmc/sq -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $O/pmc/grbm -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
cd $R
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs head -6
find $O/ubench_stats -name "*kernel_stats.csv" | head -1 | xargs head -30
ls $O/pmc/*/*/ 2>/dev/null | head
