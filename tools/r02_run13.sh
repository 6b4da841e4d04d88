#!/bin/bash
# A/B on one box: the symmetric kernel's deal of meetings to waves -- shared remainder (default) against
# whole meetings only (MAPN_SYM_PLAN=4,0,1), interleaved repetitions, three sizes
R=$PWD; O=$R/gpurun_out/r02m; rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_sym.py -m "gpu and not slow" -q 2>&1 | tail -1
for plan in 4,3 4,7 8,5 8,16 4,64 4,0,1; do MAPN_SYM_PLAN=$plan python -m pytest tests/test_gpu_sym.py -m "gpu and not slow" -q -x 2>&1 | tail -1; done
for rep in 1 2 3; do for o in 0 1; do
  MAPN_SYM_PLAN=4,0,$o python bench.py --steps 300 --warmup 20 --no-cpu-baseline > $O/bench_65536_whole${o}_$rep.json 2>/dev/null
done; done
for o in 0 1; do
  MAPN_SYM_PLAN=4,0,$o python bench.py --bodies 262144 --steps 30 --warmup 2 --no-cpu-baseline > $O/bench_262144_whole${o}.json 2>/dev/null
  MAPN_SYM_PLAN=4,0,$o python bench.py --bodies 100000 --steps 100 --warmup 5 --no-cpu-baseline > $O/bench_100000_whole${o}.json 2>/dev/null
  MAPN_SYM_PLAN=4,0,$o python bench.py --bodies 32768 --steps 300 --warmup 5 --no-cpu-baseline > $O/bench_32768_whole${o}.json 2>/dev/null
done
for p in 16 24 48 64; do MAPN_SYM_PLAN=4,$p,0 python bench.py --steps 300 --warmup 20 --no-cpu-baseline > $O/bench_65536_parts${p}.json 2>/dev/null; done
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys,os
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]); r=d.get("roofline") or {}
    print("%-30s value %.4e ms/step %.4f kernel_ms %s frac %s clk %s grid %s" % (os.path.basename(sys.argv[1]), d["value"], d["ms_per_step"], r.get("avg_launch_ms"), r.get("frac"), r.get("held_clock_ghz"), d["config"].get("grid")))
except Exception as e: print("ERR", sys.argv[1], e)
PY
done
