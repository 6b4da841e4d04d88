#!/bin/bash
# sharded symmetric step at 65 536 / 8 and / 4 in loopback: tapered parts (MAPN_SYM_SHARD_TAPER=parts,t1,t2)
for t in none 96,32,32 128,32,32 80,48,16 128,64,32 192,32,64 72,56,8; do
  if [ $t = none ]; then unset MAPN_SYM_SHARD_TAPER; else export MAPN_SYM_SHARD_TAPER=$t; fi
  echo "## taper $t"; python tools/shard_sym_loopback.py 65536 300 2>&1 | grep "symmetric  step"
done
