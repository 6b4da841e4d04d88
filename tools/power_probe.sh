#!/bin/bash
# tools/power_probe.sh -- what the package draws under each force kernel: the symmetric and then the one-sided kernel step
# 65 536 bodies flat out for 14 s while rocm-smi is read once a second from the 5th second on.  (Run ON THE GPU BOX from the
# repo root: tools/evidence.sh power.)  Finding of round 4: both kernels run AT the board's 1400 W limit; the clock a kernel
# holds is what the limit leaves it, so the rate is set by the energy a pair costs.
export MAPN_TEST_HOOKS=1
for kern in symmetric onesided; do
  python - "$kern" <<'PY' &
import os, sys, time
sys.path.insert(0, os.getcwd())
import mapn
n, kind = 65536, sys.argv[1]
c = mapn.Compute(n, device=0, mass=70000.0 / n, kernel=(mapn.KERNEL_SYMMETRIC if kind == "symmetric" else mapn.KERNEL_SCALAR))
t0, steps = time.time(), 0
while time.time() - t0 < 14:
    for _ in range(500):
        c.Simulate(n, c.GetFenceValue())
    c.WaitForGpu(); steps += 500
ms = (time.time() - t0) * 1e3 / steps
print(f"{kind}: {steps} steps, {ms:.4f} ms per step = {n * n / ms / 1e9:.3f}e12 interactions/s", flush=True)
PY
  pid=$!
  sleep 5
  for i in 1 2 3 4 5 6; do
    rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -E "Package Power|sclk|junction" | sed 's/^GPU\[0\][[:space:]]*: //' | tr '\n' ';'; echo; sleep 1
  done
  wait $pid
done
rocm-smi --showmaxpower 2>&1 | grep -i "max graphics" | sed 's/^GPU\[0\][[:space:]]*: //'
