#!/bin/bash
# Soak of the two peer-to-peer exchange forms with several REAL processes on one GPU: for each
# (world, bodies, steps) the free-running trajectory through gather algorithm 2 (exchange kernel behind
# the force launch) and through algorithm 3 (flow mode: exchange inside the force launch) must be
# bit-identical, replicas must agree on every rank (the worker asserts it) and no wait may time out.
# Every intra-GPU hand-off of flow mode is real here (slices stored write-through by one workgroup,
# read by others through the scalar cache); only the xGMI hop is not.
mkdir -p /tmp/flows
# Several processes time-slice ONE GPU here: a rank whose launch does not fit beside its peers' parks workgroups
# that starve the very peer they wait for (on a node every rank has its own GPU).  The larger jobs therefore
# run with a small forced plan (MAPN_WORKER_PLAN) so that all ranks' launches are co-resident.
for cfg in "2 8192 1500 -" "4 8192 1500 -" "8 8192 1500 -" "8 16384 400 -" "2 16384 600 sgpr,4,4,2,1" "4 16384 600 sgpr,4,4,2,1" "2 32768 200 sgpr,8,4,2,1" "4 32768 150 sgpr,8,4,2,1" "2 65536 60 sgpr,8,8,2,1"; do
  set -- $cfg; W=$1; N=$2; S=$3; PLAN=$4
  if [ "$PLAN" = "-" ]; then unset MAPN_WORKER_PLAN; else export MAPN_WORKER_PLAN=$PLAN; fi
  for mode in p2p flow; do
    rm -rf /tmp/flows/$mode; mkdir -p /tmp/flows/$mode; pids=""
    for r in $(seq 0 $((W-1))); do python tests/shard_gpu_worker.py $r $W $((29800 + W)) $N $S /tmp/flows/$mode $mode > /tmp/flows/$mode/log_$r.txt 2>&1 & pids="$pids $!"; done
    ok=1; for p in $pids; do wait $p || ok=0; done
    [ $ok = 1 ] || { echo "world=$W n=$N steps=$S mode=$mode FAILED"; for f in /tmp/flows/$mode/log_*.txt; do tail -n 2 $f; done; }
  done
  python - <<PY
import numpy as np
a=np.load("/tmp/flows/p2p/gpu_sharded.npz"); b=np.load("/tmp/flows/flow/gpu_sharded.npz")
same=all(np.array_equal(a[k],b[k]) for k in ("pos","vel","other"))
print("world=$W n=$N steps=$S plan=$PLAN  flow == p2p bitwise:", same, " finite:", bool(np.isfinite(b["pos"]).all()))
PY
done
