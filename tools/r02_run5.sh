#!/bin/bash
# round 2 evidence pass on the FINAL kernels: kernel trace + stats, the four PMC passes, shard-size
# structure timings (1-rank RCCL loopback), the other BASELINE sizes.
R=$PWD; O=$R/gpurun_out/r02e; mkdir -p $O/pmc
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc/fetch -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc/write -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/pmc/sq -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $O/pmc/grbm -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
# the same four passes for the two-kernel form (rows + reduce_integrate), for the traffic A/B
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_rows/fetch -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --plan sgpr,2,8,8,0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_rows/write -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --plan sgpr,2,8,8,0 > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $O/pmc r02 > $O/pmc_summary_stdout.txt 2>&1; tail -40 $O/pmc_summary_stdout.txt
python bench.py --steps 200 --warmup 20 > $O/bench_default.json 2> $O/bench_default.err
for n in 262144 1048576; do python bench.py --bodies $n --steps $((n==262144?60:8)) --warmup 3 --no-cpu-baseline > $O/bench_$n.json 2>&1; done
python bench.py --bodies 262144 --steps 60 --warmup 3 --no-cpu-baseline --graph > $O/bench_262144_graph.json 2>&1
python bench.py --mode central_well --bodies 4194304 --steps 400 --warmup 20 --no-cpu-baseline > $O/bench_cw_4mi.json 2>&1
{ echo "# one rank's share of the 8-way 65 536-body job (8192 x 65 536 pairs) on one GPU";
  echo "## external gather (no exchange): one launch (ticket)"; python tools/sweep.py --bodies 65536 --world 8 --auto --steps 400 --timer-interval 0;
  echo "## external gather: rows + reduce launch"; MAPN_EPILOGUE=rows python tools/sweep.py --bodies 65536 --world 8 --auto --steps 400 --timer-interval 0;
  echo "## 1-rank RCCL loopback, single launch + ncclAllGather on the compute stream"; MAPN_COMM_LOOPBACK=1 python tools/sweep.py --bodies 65536 --world 8 --auto --steps 400 --timer-interval 0;
  echo "## 1-rank RCCL loopback, own/remote overlap structure (two launches sharing the tickets)"; MAPN_COMM_LOOPBACK=1 python tools/sweep.py --bodies 65536 --world 8 --auto --overlap --steps 400 --timer-interval 0;
  echo "# one rank's share of the 8-way 1 048 576-body job (131 072 x 1 048 576 pairs)";
  python tools/sweep.py --bodies 1048576 --world 8 --auto --steps 10 --timer-interval 0; } > $O/shard_structure.txt 2>&1
cat $O/shard_structure.txt
for f in $O/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    r=d.get("roofline") or {}
    print(d["value"], d["ms_per_step"], r.get("avg_launch_ms"), r.get("frac"), r.get("held_clock_ghz"), r.get("frac_at_held_clock"), r.get("traffic"))
except Exception as e: print("ERR",e)
PY
done
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs head -6
