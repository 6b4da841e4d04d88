#!/bin/bash
R=$PWD; O=$R/gpurun_out/r02h; mkdir -p $O
python -m pytest tests/test_gpu_sym.py -m gpu -q -x > $O/pytest_sym.txt 2>&1; tail -2 $O/pytest_sym.txt
for plan in 8,16 8,24 8,32; do
  MAPN_SYM_PLAN=$plan python bench.py --kernel sym --steps 200 --warmup 20 --no-cpu-baseline > $O/b.json 2> $O/b.err
  python - "$O/b.json" "$plan" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]); r=d["roofline"]
    print("plan", sys.argv[2] or "auto", "value %.4e ms/step %.4f kernel ms %.4f grid %s block %s" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], d["config"]["grid"], d["config"]["block"]))
except Exception as e: print("ERR", sys.argv[2], e)
PY
done
for n in 262144 1048576; do MAPN_SYM_MAX_MB=20000 python bench.py --kernel sym --bodies $n --steps $((n==262144?40:6)) --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('sym N=%d value %.4e ms/step %.3f grid %s' % (d['config']['bodies'], d['value'], d['ms_per_step'], d['config']['grid']))"; done
python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('one-sided default: value %.4e ms/step %.4f' % (d['value'], d['ms_per_step']))"
