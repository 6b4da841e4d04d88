#!/bin/bash
# the GPU test suite (optionally without the slow oracle legs) + the default bench line
R=$PWD; O=$R/gpurun_out/r03f; mkdir -p $O
python -m pytest tests -m "gpu and not slow" -x -q > $O/pytest_gpu_fast.txt 2>&1; tail -15 $O/pytest_gpu_fast.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json
