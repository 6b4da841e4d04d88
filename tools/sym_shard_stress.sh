#!/bin/bash
# Soak of the sharded symmetric step (gather algorithm 4) with several REAL processes on one GPU: for each
# (world, bodies, steps) the free-running trajectory must stay with the one-sided sharded step's (algorithm 2)
# like two summation orders of the same dynamics do, replicas must agree on every rank and no wait may time
# out (the worker asserts both).  Every intra-GPU hand-off is real here (reaction rows stored write-through into
# another process's receive region through the hipIpc mapping, flags, bounded waits); only the xGMI hop is not.
mkdir -p /tmp/syms
for cfg in "2 8192 1500" "4 8192 1500" "8 8192 1500" "3 9216 800" "8 16384 600" "4 32768 300" "2 65536 120" "8 65536 120" "8 262144 12"; do
  set -- $cfg; W=$1; N=$2; S=$3
  for mode in p2p sym; do
    rm -rf /tmp/syms/$mode; mkdir -p /tmp/syms/$mode; pids=""
    for r in $(seq 0 $((W-1))); do python tests/shard_gpu_worker.py $r $W $((29850 + W)) $N $S /tmp/syms/$mode $mode > /tmp/syms/$mode/log_$r.txt 2>&1 & pids="$pids $!"; done
    ok=1; for p in $pids; do wait $p || ok=0; done
    [ $ok = 1 ] || { echo "world=$W n=$N steps=$S mode=$mode FAILED"; for f in /tmp/syms/$mode/log_*.txt; do tail -n 2 $f; done; }
  done
  python - <<PY
import numpy as np
a=np.load("/tmp/syms/p2p/gpu_sharded.npz"); b=np.load("/tmp/syms/sym/gpu_sharded.npz")
d=np.linalg.norm(a["pos"][:,:3].astype(np.float64)-b["pos"][:,:3],axis=1)/np.maximum(np.linalg.norm(a["pos"][:,:3].astype(np.float64),axis=1),1e-30)
print("world=$W n=$N steps=$S  symmetric vs one-sided sharded step: relative position difference max %.2e median %.2e" % (d.max(), np.median(d)), " finite:", bool(np.isfinite(b["pos"]).all()))
PY
done
# yardstick for the long small runs: the same comparison between the UNSHARDED symmetric and one-sided kernels
python - <<'PY'
import numpy as np, mapn
for n, steps in ((8192, 1500), (9216, 800), (16384, 600)):
    out = []
    for k in (mapn.KERNEL_SYMMETRIC, mapn.KERNEL_SCALAR):
        with mapn.Compute(n, mass=70000.0 / n, kernel=k) as c:
            for _ in range(steps):
                c.Simulate(n, c.GetFenceValue())
            out.append(c.download_state()[0][:, :3].astype(np.float64))
    d = np.linalg.norm(out[0] - out[1], axis=1) / np.maximum(np.linalg.norm(out[1], axis=1), 1e-30)
    print("one GPU  n=%d steps=%d  symmetric vs one-sided kernel: relative position difference max %.2e median %.2e" % (n, steps, d.max(), np.median(d)))
PY
