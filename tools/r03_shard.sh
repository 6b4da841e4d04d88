#!/bin/bash
# sharded symmetric step: multi-process tests (algorithms 4 and 5), loopback timelines of 4 / 5
R=$PWD; O=$R/gpurun_out/r03g; rm -rf $O; mkdir -p $O
python -m pytest tests/test_shard_gpu_multiproc.py -x -q > $O/pytest_shard.txt 2>&1; tail -8 $O/pytest_shard.txt
for algo in 4 5; do python tools/shard_timeline.py 65536 8 0 $algo > $O/timeline_rank0_algo$algo.txt 2>&1; head -2 $O/timeline_rank0_algo$algo.txt; done
python tools/shard_timeline.py 65536 8 4 5 > $O/timeline_rank4_algo5.txt 2>&1; head -2 $O/timeline_rank4_algo5.txt
MAPN_SYM_SHARD_PLAN=8,32 python tools/shard_timeline.py 65536 8 0 5 > $O/timeline_rank0_algo5_w8.txt 2>&1; head -2 $O/timeline_rank0_algo5_w8.txt
