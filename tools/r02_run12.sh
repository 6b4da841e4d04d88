#!/bin/bash
# A/B of bodies-per-lane in the symmetric kernel on one box: K2 = 8 (16 bodies, 2 waves/SIMD, shipped),
# (the variant template parameter / -DMAPN_SYM_K2 builds existed only in the experiment commits; see profiles/r02_sym_loop_variants.txt)
# 6 (12 bodies, 3 waves/SIMD), 4 (8 bodies, 4 waves/SIMD); experiment libraries built with -DMAPN_SYM_K2.
R=$PWD; O=$R/gpurun_out/r02l; mkdir -p $O
L=multi-adapter-particles_amd
cp $L/libmapn.so $L/libmapn_k8.so
for k in 8 4 6; do
  cp $L/libmapn_k$k.so $L/libmapn.so
  MAPN_SYM_PLAN=4,0,1 python -m pytest tests/test_gpu_sym.py -m "gpu and not slow" -q 2>&1 | tail -1
  for v in 1 2; do
    MAPN_SYM_PLAN=4,0,$v python bench.py --steps 300 --warmup 20 --no-cpu-baseline > $O/bench_k${k}_v${v}.json 2>/dev/null
    MAPN_SYM_PLAN=4,0,$v python bench.py --bodies 262144 --steps 30 --warmup 2 --no-cpu-baseline > $O/bench_262144_k${k}_v${v}.json 2>/dev/null
  done
done
cp $L/libmapn_k8.so $L/libmapn.so
MAPN_SYM_PLAN=4,0,1 python bench.py --steps 300 --warmup 20 --no-cpu-baseline > $O/bench_k8_v1_again.json 2>/dev/null
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys,os
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]); r=d.get("roofline") or {}
    print("%-30s value %.4e ms/step %.4f kernel_ms %s frac %s clk %s grid %s" % (os.path.basename(sys.argv[1]), d["value"], d["ms_per_step"], r.get("avg_launch_ms"), r.get("frac"), r.get("held_clock_ghz"), d["config"].get("grid")))
except Exception as e: print("ERR", sys.argv[1], e)
PY
done
