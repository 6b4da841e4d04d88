#!/bin/bash
# re-key the PMC summaries after an edit of the reduce kernel (a-row loads batched): four --pmc passes per force kernel,
# kernel trace + stats of the default bench, the default bench line
R=$PWD; O=$R/gpurun_out/r02x; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for k in sym sgpr; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_$k/fetch -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --kernel $k > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_$k/write -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --kernel $k > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_$k/sq -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --kernel $k > /dev/null 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $O/pmc_$k/grbm -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --kernel $k > /dev/null 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sym -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_profiled_sym.json 2> $O/bench_profiled_sym.err
cd $R
python tools/pmc_summary.py $O/pmc_sym r02_sym > $O/pmc_summary_sym.txt 2>&1
python tools/pmc_summary.py $O/pmc_sgpr r02_onesided > $O/pmc_summary_onesided.txt 2>&1
cp profiles/r02_sym_pmc_summary.* profiles/r02_onesided_pmc_summary.* $O/
f=$(ls -t $(find $O/stats_sym -name "*kernel_stats.csv") | head -1); cp $f $O/kernel_stats_sym.csv; head -4 $f
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
for f in ("bench_default","bench_profiled_sym"):
    d=json.loads([l for l in open(f"gpurun_out/r02x/{f}.json").read().splitlines() if l.startswith("{")][-1]); r=d["roofline"]
    print(f, "value %.4e ms %.4f kernel_ms %.4f frac %.3f clk %s traffic %s" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], r["held_clock_ghz"], r["traffic"]))
PY
tail -1 $O/pmc_summary_sym.txt
