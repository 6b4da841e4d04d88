#!/bin/bash
# refresh of the sha-keyed PMC summaries after comment-only edits of the kernel sources (same kernels as
# tools/r02_run17.sh), the default bench line with the PMC traffic attached, and the soak of the sharded symmetric step
R=$PWD; O=$R/gpurun_out/r02s; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for k in sym sgpr; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_$k/fetch -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --kernel $k > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_$k/write -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --kernel $k > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_$k/sq -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --kernel $k > /dev/null 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $O/pmc_$k/grbm -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --kernel $k > /dev/null 2>&1
done
cd $R
python tools/pmc_summary.py $O/pmc_sym r02_sym > $O/pmc_summary_sym.txt 2>&1
python tools/pmc_summary.py $O/pmc_sgpr r02_onesided > $O/pmc_summary_onesided.txt 2>&1
cp profiles/r02_sym_pmc_summary.* profiles/r02_onesided_pmc_summary.* $O/
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2>/dev/null
python - <<'PY'
import json
for f in ("bench_default","bench_driver_form"):
    d=json.loads([l for l in open(f"gpurun_out/r02s/{f}.json").read().splitlines() if l.startswith("{")][-1]); r=d["roofline"]
    print(f, "value %.4e ms %.4f frac %.3f clk %s traffic %s cpu %s" % (d["value"], d["ms_per_step"], r["frac"], r["held_clock_ghz"], r["traffic"], (d.get("cpu_baseline") or {}).get("value")))
PY
tail -2 $O/pmc_summary_sym.txt
bash tools/sym_shard_stress.sh 2>&1 | tee $O/sym_shard_soak.txt | tail -14
