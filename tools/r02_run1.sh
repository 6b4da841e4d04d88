#!/bin/bash
# round 2, first GPU pass: full -m gpu suite, ticket-vs-rows A/B of the default step, kernel trace.
R=$PWD; O=$R/gpurun_out/r02a; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -5 $O/pytest_gpu.txt
for rep in 1 2; do
  for plan in sgpr,2,8,8,1 sgpr,2,8,8,0; do
    python bench.py --steps 400 --warmup 20 --no-cpu-baseline --plan $plan > $O/bench_${plan//,/_}_$rep.json 2> $O/bench_${plan//,/_}_$rep.err
  done
done
python bench.py --steps 200 --warmup 20 > $O/bench_default.json 2> $O/bench_default.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
cd $R
for f in $O/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d.get("roofline",{}).get("avg_launch_ms"), d.get("roofline",{}).get("frac"), d["config"]["fused_integrator"])
except Exception as e: print("ERR",e)
PY
done
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs head -8
