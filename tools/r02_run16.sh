#!/bin/bash
# the sharded symmetric step (gather algorithm 4): correctness between processes, then rank 0's compute at the
# true shard size in loopback (one-sided against symmetric), workgroups per I-block swept at the 8-way size
R=$PWD; O=$R/gpurun_out/r02p; mkdir -p $O
timeout 900 python -m pytest tests/test_shard_gpu_multiproc.py -q -k symmetric 2>&1 | tail -2
python tools/shard_sym_loopback.py 65536 400 2>&1 | tee $O/loopback_65536.txt
for p in 32 48 96 128; do echo "## MAPN_SYM_SHARD_PARTS=$p"; MAPN_SYM_SHARD_PARTS=$p python tools/shard_sym_loopback.py 65536 400 2>&1 | grep "world 8  symmetric "; done | tee $O/loopback_65536_parts.txt
python tools/shard_sym_loopback.py 262144 40 2>&1 | tee $O/loopback_262144.txt
python tools/shard_sym_loopback.py 1048576 6 2>&1 | tee $O/loopback_1048576.txt
