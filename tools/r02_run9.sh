#!/bin/bash
R=$PWD; O=$R/gpurun_out/r02i; mkdir -p $O
python -m pytest tests -m "gpu and not slow" -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -30 $O/pytest_gpu.txt
python bench.py --steps 200 --warmup 20 > $O/bench_default.json 2> $O/bench_default.err; cut -c1-3000 $O/bench_default.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_like.json 2>/dev/null; cut -c1-200 $O/bench_driver_like.json
python -m pytest tests/test_parity_1000.py -m gpu -q -s > $O/pytest_parity1000.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_parity1000.txt
grep -E "passed|failed|vs ref @1000|@1000 vs acc64|# oracle" $O/pytest_parity1000.txt | cut -c1-400
