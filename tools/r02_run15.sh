#!/bin/bash
# symmetric kernel, 65 536 bodies: workgroups per I-block (parts) with the shared-remainder deal, one box
R=$PWD; O=$R/gpurun_out/r02o; rm -rf $O; mkdir -p $O
for rep in 1 2; do for p in 8 12 16 32; do
  MAPN_SYM_PLAN=4,$p python bench.py --steps 300 --warmup 20 --no-cpu-baseline > $O/bench_65536_parts${p}_$rep.json 2>/dev/null
done; done
for p in 4 8 16 32; do MAPN_SYM_PLAN=4,$p python bench.py --bodies 262144 --steps 30 --warmup 2 --no-cpu-baseline > $O/bench_262144_parts${p}.json 2>/dev/null; done
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys,os
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]); r=d.get("roofline") or {}
    print("%-32s value %.4e ms/step %.4f kernel_ms %.4f frac %.3f clk %.3f grid %s" % (os.path.basename(sys.argv[1]), d["value"], d["ms_per_step"], r.get("avg_launch_ms"), r.get("frac"), r.get("held_clock_ghz"), d["config"].get("grid")))
except Exception as e: print("ERR", sys.argv[1], e)
PY
done
