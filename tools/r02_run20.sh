#!/bin/bash
# tapered parts of the symmetric kernel (small workgroups last): same-box A/B against equal parts, several tapers and sizes
R=$PWD; O=$R/gpurun_out/r02v; rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_sym.py -m "gpu and not slow" -q 2>&1 | tail -1
for t in 48,24,8 40,28,4 64,16,16 36,28,8; do MAPN_SYM_TAPER=$t python -m pytest tests/test_gpu_sym.py -m "gpu and not slow" -q -x 2>&1 | tail -1; done
for rep in 1 2 3; do
  for t in 0 48,24,8 40,28,4 56,20,8 64,16,16 36,28,8 72,24,24; do
    MAPN_SYM_TAPER=$t python bench.py --steps 300 --warmup 20 --no-cpu-baseline > $O/bench_65536_t${t//,/_}_$rep.json 2>/dev/null
  done
done
for n in 32768 100000 131072 262144; do for t in 0 48,24,8; do
  MAPN_SYM_TAPER=$t python bench.py --bodies $n --steps $((n>=262144?40:200)) --warmup 5 --no-cpu-baseline > $O/bench_${n}_t${t//,/_}.json 2>/dev/null
done; done
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys,os
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]); r=d.get("roofline") or {}
    print("%-36s value %.4e ms/step %.4f kernel_ms %.4f frac %.3f clk %.3f grid %s" % (os.path.basename(sys.argv[1]), d["value"], d["ms_per_step"], r.get("avg_launch_ms"), r.get("frac"), r.get("held_clock_ghz"), d["config"].get("grid")))
except Exception as e: print("ERR", sys.argv[1], e)
PY
done
