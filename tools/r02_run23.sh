#!/bin/bash
# how much of a SIMD's VALU can ONE resident wave of force_sym_kernel use?  (one workgroup per CU via LDS padding)
for pad in 0 60000; do for n in 65536 262144; do
  echo "## MAPN_SYM_PAD_LDS=$pad bodies $n"
  MAPN_SYM_PAD_LDS=$pad MAPN_SYM_TAPER=0 python bench.py --bodies $n --steps $((n>65536?20:100)) --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r=d['roofline']
print('value %.4e kernel_ms %.4f clk %.3f' % (d['value'], r['avg_launch_ms'], r['held_clock_ghz']))"
done; done
