#!/bin/bash
# round 2 evidence pass on the FINAL kernels (16-byte row stores): suite, kernel trace + stats, four PMC passes, A/B.
R=$PWD; O=$R/gpurun_out/r02f; mkdir -p $O/pmc
python -m pytest tests -m "gpu and not slow" -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -6 $O/pytest_gpu.txt
for rep in 1 2; do
  python bench.py --steps 400 --warmup 20 --no-cpu-baseline > $O/bench_ticket_$rep.json 2>/dev/null
  python bench.py --steps 400 --warmup 20 --no-cpu-baseline --plan sgpr,2,8,8,0 > $O/bench_rows_$rep.json 2>/dev/null
done
python bench.py --steps 200 --warmup 20 > $O/bench_default.json 2> $O/bench_default.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc/fetch -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc/write -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/pmc/sq -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $O/pmc/grbm -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $O/pmc r02 > $O/pmc_summary_stdout.txt 2>&1; tail -26 $O/pmc_summary_stdout.txt
cp profiles/r02_pmc_summary.* $O/
for f in $O/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    r=d.get("roofline") or {}
    print(d["value"], d["ms_per_step"], r.get("avg_launch_ms"), r.get("frac"), r.get("held_clock_ghz"), r.get("frac_at_held_clock"), r.get("traffic"))
except Exception as e: print("ERR",e)
PY
done
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs head -4
