#!/bin/bash
# A/B of the symmetric kernel's inner-loop variants on one box (MAPN_SYM_PLAN=waves,parts,variant):
# (the variant template parameter / -DMAPN_SYM_K2 builds existed only in the experiment commits; see profiles/r02_sym_loop_variants.txt)
# 0 = reaction folded every step (6 moves + 3 adds + copies), 1 = reaction travels unfolded (9 moves),
# 2 = 1 unrolled by two (no loop-carried copies), 3 = 2 with the position moves pinned at the step head.
R=$PWD; O=$R/gpurun_out/r02k; mkdir -p $O
for v in 1 2 3; do MAPN_SYM_PLAN=4,0,$v python -m pytest tests/test_gpu_sym.py -m "gpu and not slow" -q -x 2>&1 | tail -1; done
for rep in 1 2; do for v in 0 1 2 3; do
  MAPN_SYM_PLAN=4,0,$v python bench.py --steps 300 --warmup 20 --no-cpu-baseline > $O/bench_v${v}_$rep.json 2>/dev/null
done; done
for v in 0 1 2 3; do MAPN_SYM_PLAN=4,0,$v python bench.py --bodies 262144 --steps 30 --warmup 2 --no-cpu-baseline > $O/bench_262144_v${v}.json 2>/dev/null; done
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys,os
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]); r=d.get("roofline") or {}
    print("%-30s value %.4e ms/step %.4f kernel_ms %s frac %s clk %s" % (os.path.basename(sys.argv[1]), d["value"], d["ms_per_step"], r.get("avg_launch_ms"), r.get("frac"), r.get("held_clock_ghz")))
except Exception as e: print("ERR", sys.argv[1], e)
PY
done
