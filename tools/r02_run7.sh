#!/bin/bash
R=$PWD; O=$R/gpurun_out/r02g; mkdir -p $O
python -m pytest tests/test_gpu_sym.py -m gpu -q -x -s > $O/pytest_sym.txt 2>&1; echo "rc=$?" >> $O/pytest_sym.txt; tail -15 $O/pytest_sym.txt
for plan in "" 8,4 8,8 8,16 4,8 4,16 8,2; do
  MAPN_SYM_PLAN=$plan python bench.py --kernel sym --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_sym_${plan//,/_}.json 2> $O/bench_sym_${plan//,/_}.err
  python - "$O/bench_sym_${plan//,/_}.json" "$plan" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]); r=d["roofline"]
    print("plan", sys.argv[2] or "auto", "value %.4e ms/step %.4f kernel ms %.4f grid %s block %s" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], d["config"]["grid"], d["config"]["block"]))
except Exception as e: print("ERR", sys.argv[2], e, open(sys.argv[1].replace(".json",".err")).read()[-400:])
PY
done
python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('one-sided default: value %.4e ms/step %.4f' % (d['value'], d['ms_per_step']))"
