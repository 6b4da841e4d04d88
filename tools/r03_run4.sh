#!/bin/bash
R=$PWD; O=$R/gpurun_out/r03i; rm -rf $O; mkdir -p $O
for ap in 0 2 3 4 6; do echo "== alt_prio $ap"; MAPN_SYM_ALT_PRIO=$ap python tools/shard_timeline.py 65536 8 0 5 2>&1 | head -2; done | tee $O/alt_prio.txt
for ap in 0 3 0 3; do MAPN_SYM_ALT_PRIO=$ap python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('alt_prio $ap', 'ms/step', round(d['ms_per_step'],4), 'force ms', round(d['roofline']['avg_launch_ms'],4), 'clk', d['roofline'].get('held_clock_ghz'))"; done | tee -a $O/alt_prio.txt
python -m pytest tests/test_bench_contract.py -m gpu -x -q -k eight > $O/pytest_bench8.txt 2>&1; tail -5 $O/pytest_bench8.txt; grep -o "\[bench\].*" $O/pytest_bench8.txt | cut -c1-400 | tail -8
