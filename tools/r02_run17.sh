#!/bin/bash
# round 2 final evidence pass (supersedes r02_run10.sh after the symmetric kernel's loop / deal changes and the
# sharded symmetric step): tests, both force kernels under rocprofv3 (trace + stats, four PMC passes each),
# other sizes, sharded steps at the true shard size in loopback, 1000-step parity.
R=$PWD; O=$R/gpurun_out/r02q; rm -rf $O; mkdir -p $O
python -m pytest tests -m "gpu and not slow" -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt
for rep in 1 2; do
  python bench.py --steps 400 --warmup 20 --no-cpu-baseline > $O/bench_sym_$rep.json 2>/dev/null
  python bench.py --steps 400 --warmup 20 --no-cpu-baseline --kernel sgpr > $O/bench_onesided_$rep.json 2>/dev/null
done
python bench.py --steps 200 --warmup 20 > $O/bench_default.json 2> $O/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2>/dev/null
cd /tmp; export TMPDIR=/tmp
for k in sym sgpr; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$k -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --kernel $k > $O/bench_profiled_$k.json 2> $O/bench_profiled_$k.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_$k/fetch -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --kernel $k > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_$k/write -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --kernel $k > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_$k/sq -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --kernel $k > /dev/null 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $O/pmc_$k/grbm -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --kernel $k > /dev/null 2>&1
done
cd $R
python tools/pmc_summary.py $O/pmc_sym r02_sym > $O/pmc_summary_sym.txt 2>&1
python tools/pmc_summary.py $O/pmc_sgpr r02_onesided > $O/pmc_summary_onesided.txt 2>&1
cp profiles/r02_sym_pmc_summary.* profiles/r02_onesided_pmc_summary.* $O/
for k in sym sgpr; do f=$(ls -t $(find $O/stats_$k -name "*kernel_stats.csv") | head -1); cp $f $O/kernel_stats_$k.csv; head -4 $f; done
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_default_with_new_pmc.json 2>/dev/null
for n in 32768 100000 262144 1048576; do
  for k in sym sgpr; do MAPN_SYM_MAX_MB=20000 python bench.py --kernel $k --bodies $n --steps $((n>=1048576?6:(n>=262144?40:200))) --warmup 2 --no-cpu-baseline > $O/bench_${n}_$k.json 2>/dev/null; done
done
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys,os
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]); r=d.get("roofline") or {}
    print("%-40s value %.4e ms/step %.4f kernel_ms %s frac %s clk %s grid %s traffic %s" % (os.path.basename(sys.argv[1]), d["value"], d["ms_per_step"], r.get("avg_launch_ms"), r.get("frac"), r.get("held_clock_ghz"), d["config"]["grid"], r.get("traffic")))
except Exception as e: print("ERR", sys.argv[1], e)
PY
done | tee $O/bench_table.txt
python tools/shard_sym_loopback.py 65536 400 2>&1 | tee $O/loopback_65536.txt
python tools/shard_sym_loopback.py 262144 40 2>&1 | tee $O/loopback_262144.txt
python tools/shard_sym_loopback.py 1048576 6 2>&1 | tee $O/loopback_1048576.txt
python -m pytest tests/test_parity_1000.py -m gpu -q -s > $O/pytest_parity1000.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_parity1000.txt
grep -E "passed|failed|vs ref @1000|@1000 vs acc64|# oracle" $O/pytest_parity1000.txt | cut -c1-300
cp profiles/r02_parity_1000_65536_*.json $O/ 2>/dev/null
