#!/bin/bash
# experiment: time-sliced issue priority between the two waves of a SIMD (MAPN_SYM_FAIR=log2 of the slice in cycles)
# (the MAPN_SYM_FAIR hook existed only in the experiment build; it was removed after this measurement)
python -m pytest tests/test_gpu_sym.py -m "gpu and not slow" -q 2>&1 | tail -1
MAPN_SYM_FAIR=13 python -m pytest tests/test_gpu_sym.py -m "gpu and not slow" -q 2>&1 | tail -1
for rep in 1 2; do for f in 0 11 13 15 17; do
  echo -n "fair=$f unsharded 65536: "; MAPN_SYM_FAIR=$f python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r=d['roofline']
print('value %.4e kernel_ms %.4f clk %.3f' % (d['value'], r['avg_launch_ms'], r['held_clock_ghz']))"
done; done
for f in 0 11 13 15 17; do echo "## fair=$f sharded loopback"; MAPN_SYM_FAIR=$f python tools/shard_sym_loopback.py 65536 300 2>&1 | grep "symmetric  step"; done
echo -n "fair=13 one round (8 equal parts): "; MAPN_SYM_FAIR=13 MAPN_SYM_PLAN=4,8 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r=d['roofline']
print('value %.4e kernel_ms %.4f' % (d['value'], r['avg_launch_ms']))"
echo -n "fair=0 one round (8 equal parts): "; MAPN_SYM_PLAN=4,8 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r=d['roofline']
print('value %.4e kernel_ms %.4f' % (d['value'], r['avg_launch_ms']))"
