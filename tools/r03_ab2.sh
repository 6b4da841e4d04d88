#!/bin/bash
# same-box A/B after the vmcnt fix: round 2's kernels (ab/old) vs the working tree; row write-through and 8-wave workgroups on the sharded launch
R=$PWD; O=$R/gpurun_out/r03e; rm -rf $O; mkdir -p $O
B() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'ms/step', round(d['ms_per_step'],4), 'force ms', round(d['roofline']['avg_launch_ms'],4), 'clk', d['roofline'].get('held_clock_ghz'))"; }
for rep in 1 2; do
  (cd ab/old && python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | B old) >> $O/bench.txt
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | B new >> $O/bench.txt
  MAPN_SYM_ROW_WT=1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | B new_wt >> $O/bench.txt
done
cat $O/bench.txt
(cd ab/old && python tools/shard_sym_loopback.py 65536 400 2>&1 | grep "world 8  sym") > $O/loop.txt
L() { echo "== $1" >> $O/loop.txt; python tools/shard_timeline.py 65536 8 0 4 2>&1 | head -2 >> $O/loop.txt; }
L default
MAPN_SYM_ROW_WT=1 L row_wt
MAPN_SYM_SHARD_PLAN=8,32 L waves8
MAPN_SYM_ROW_WT=1 MAPN_SYM_SHARD_PLAN=8,32 L waves8_row_wt
MAPN_SYM_SHARD_PULL=0 L nopull
MAPN_SYM_SHARD_PULL=0 MAPN_SYM_ROW_WT=1 L nopull_row_wt
cat $O/loop.txt
