#!/bin/bash
# same-box A/B: round 2's symmetric kernels (git HEAD of round 2, exported to ab/old) against the working tree,
# rank 0's loopback step at 65 536 / 8 and the unsharded step
R=$PWD; O=$R/gpurun_out/r03c; rm -rf $O; mkdir -p $O
for rep in 1 2; do
  (cd ab/old && python tools/shard_sym_loopback.py 65536 400 2>&1 | grep "world 8") >> $O/old.txt
  python tools/shard_sym_loopback.py 65536 400 2>&1 | grep "world 8" >> $O/new.txt
  (cd ab/old && python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('old bench', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline'].get('held_clock_ghz'))") >> $O/old.txt
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new bench', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline'].get('held_clock_ghz'))" >> $O/new.txt
done
echo OLD; cat $O/old.txt; echo NEW; cat $O/new.txt
