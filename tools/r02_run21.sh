#!/bin/bash
# confirm the default taper (40 parts: 28 x 4, 4 x 2, 8 x 1 units) against equal parts, interleaved, one box
R=$PWD; O=$R/gpurun_out/r02w; rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_sym.py tests/test_gpu_parity.py -m "gpu and not slow" -q 2>&1 | tail -1
for rep in 1 2 3; do for t in 0 d; do
  if [ $t = d ]; then unset MAPN_SYM_TAPER; else export MAPN_SYM_TAPER=0; fi
  python bench.py --steps 300 --warmup 20 --no-cpu-baseline > $O/bench_65536_t${t}_$rep.json 2>/dev/null
done; done
for n in 100000 131072; do for t in 0 d; do
  if [ $t = d ]; then unset MAPN_SYM_TAPER; else export MAPN_SYM_TAPER=0; fi
  python bench.py --bodies $n --steps 200 --warmup 5 --no-cpu-baseline > $O/bench_${n}_t${t}.json 2>/dev/null
done; done
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys,os
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]); r=d.get("roofline") or {}
print("%-28s value %.4e ms/step %.4f kernel_ms %.4f frac %.3f clk %.3f grid %s" % (os.path.basename(sys.argv[1]), d["value"], d["ms_per_step"], r.get("avg_launch_ms"), r.get("frac"), r.get("held_clock_ghz"), d["config"].get("grid")))
PY
done
