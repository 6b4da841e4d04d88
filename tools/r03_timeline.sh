#!/bin/bash
# round 3, step 1: where does the sharded symmetric force launch at 65 536 / 8 lose its time?  Per-wave timelines (rank 0: 528
# meetings per block, rank 4: 512), the unsharded launch for comparison.
R=$PWD; O=$R/gpurun_out/r03a; rm -rf $O; mkdir -p $O
python tools/shard_timeline.py 65536 8 0 4 > $O/timeline_rank0.txt 2>&1
python tools/shard_timeline.py 65536 8 4 4 > $O/timeline_rank4.txt 2>&1
python tools/shard_timeline.py 65536 1 0 0 > $O/timeline_unsharded.txt 2>&1
python tools/shard_timeline.py 262144 8 0 4 > $O/timeline_262144_rank0.txt 2>&1
cat $O/timeline_rank0.txt $O/timeline_rank4.txt
