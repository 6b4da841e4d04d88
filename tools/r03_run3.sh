#!/bin/bash
R=$PWD; O=$R/gpurun_out/r03h; rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_sym.py -x -q -s > $O/pytest_sym.txt 2>&1; grep -E "passed|failed|order-matched|windows" $O/pytest_sym.txt | tail -20
python -m pytest tests/test_bench_contract.py -m gpu -x -q > $O/pytest_bench.txt 2>&1; tail -5 $O/pytest_bench.txt
for st in 0 1; do MAPN_SYM_STAGE=$st python tools/shard_timeline.py 65536 8 0 5 > $O/timeline_stage$st.txt 2>&1; head -8 $O/timeline_stage$st.txt; done
for st in 0 1 0 1; do MAPN_SYM_STAGE=$st python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stage $st', 'ms/step', round(d['ms_per_step'],4), 'force ms', round(d['roofline']['avg_launch_ms'],4), 'clk', d['roofline'].get('held_clock_ghz'), d['config']['step_ms_by_quarter_of_the_timed_region'])"; done
