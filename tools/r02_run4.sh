#!/bin/bash
R=$PWD; O=$R/gpurun_out/r02d; mkdir -p $O
python -m pytest tests -m "gpu and not slow" -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -25 $O/pytest_gpu.txt
python bench.py --steps 200 --warmup 20 > $O/bench_default.json 2> $O/bench_default.err; cut -c1-2600 $O/bench_default.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_like.json 2> $O/bench_driver_like.err; cut -c1-300 $O/bench_driver_like.json
python bench.py --steps 400 --warmup 20 --no-cpu-baseline --plan sgpr,2,8,8,0 > $O/bench_rows.json 2>&1; cut -c1-300 $O/bench_rows.json
