#!/bin/bash
# tools/evidence.sh <what> [args]  -- every measurement behind profiles/ and DESIGN.md, run ON THE GPU BOX from the repo root
# (gpurun -- 'bash tools/evidence.sh <what> ...').  Output goes to gpurun_out/<tag>/ (scratch); what is to be judged is copied
# into profiles/ by hand.  One script with parameters instead of a numbered script per lease.
#
#   suite [fast|full]            pytest -m gpu (fast: without the slow oracle legs) + the default bench line
#   bench [bench.py args]        one bench line, summarised
#   stats [sym|sgpr]             rocprofv3 --kernel-trace --stats of the default bench -> kernel_stats csv
#   pmc   [sym|sgpr] <tag>       the four --pmc passes of one force kernel -> profiles/<tag>_pmc_summary.{txt,json} (tools/pmc_summary.py)
#   issue [tree]                 SQ issue / wait counters of force_sym_kernel (three --pmc passes)
#   ab <other-tree>              same-box A/B of another checkout (e.g. ab/old: `git archive <rev> | tar -x -C ab/old && make -C ...`)
#   timeline N WORLD RANK ALGO   per-wave timeline of one sharded symmetric force launch in loopback (tools/shard_timeline.py)
#   shardstep [N WORLD [TREE]]   kernel trace of the loopback step, ranks 0 and WORLD/2, algorithms 4 and 5: durations, gaps, period
#   shardpartial [N WORLD]       the same for PARTIALLY ACTIVE steps (num_active = N/2, 5N/8): the split form (algorithm 5) on an active and on a frozen
#                                rank against algorithm 2's step (one-sided kernel over the rank's active bodies x N + pull) -> shard_partial_timeline.txt
#   loopback [N]                 rank 0's compute per step at the shard size, one-sided against symmetric, WORLD = 2, 4, 8
#   sizes                        symmetric against one-sided kernel at 65 536 ... 4 194 304 bodies
#   parity1000                   tests/parity_report.py at 65 536 bodies, 1000 steps, all legs -> JSON
#   ubench                       BASELINE configs[4]: the MFMA-against-packed-VALU A/B (tools/ubench --ab) under rocprofv3 --kernel-trace --stats + the gpu test's report
#   soak MODE                    several real processes on one GPU, long runs: MODE = flow | sym | sympush (against p2p)
#   slider                       ms per step with num_active steady / alternating between two counts / NEW every step (the slider dragged), one GPU (tools/slider_drag.py)
#   soakslider                   the slider dragged on a sharded job for many steps (partially active steps in their split form, all-active, one-sided, nothing), several real
#                                processes on one GPU: pushed (5) against pulled (4) positions bit for bit -> slider_soak.txt
#   closing                      what the ONE closing collective (barrier + verdict) adds to a timed region of K = 20 / 200 steps: two ranks sharing this GPU
#                                (gloo) and one rank over the RCCL backend -> closing_cost.txt
#   partial                      partially active steps: one-sided / full symmetric / split form over a sweep of num_active (tools/partial_sweep.py)
#   partialstats [N ACTIVE]      rocprofv3 --kernel-trace --stats of 2000 half-active steps in the split form: the three launches' durations
#   rankfail                     eight real processes on one GPU (gloo): rank 1 corrupts one pushed position (a) while the exchange TRIAL steps that form, (b) in
#                                the timed region -- what every rank decides, in bench.py's own words -> rank_failure_8_ranks.txt
#   power                        rocm-smi package power / shader clock / temperature while each force kernel runs flat out for 14 s (tools/power_probe.sh)
set -u
R=$PWD; W=${1:-suite}; shift || true
O=$R/gpurun_out/ev_$W; mkdir -p $O
export TMPDIR=/tmp
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r=d.get('roofline') or {}; print('$1', 'value %.4e' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'force ms', round(r.get('avg_launch_ms',0),4), 'frac', round(r.get('frac',0),4), 'executed', round(r.get('frac_executed',0),4), 'clk', r.get('held_clock_ghz'), 'quarters', d['config'].get('step_ms_by_quarter_of_the_timed_region'), d['config'].get('kernel'))"; }
case $W in
suite)
  sel=${1:-fast}; [ $sel = fast ] && M="gpu and not slow" || M="gpu"
  python -m pytest tests -m "$M" -q > $O/pytest_$sel.txt 2>&1; tail -4 $O/pytest_$sel.txt
  python bench.py > $O/bench_default.json 2> $O/bench_default.err; line default < $O/bench_default.json ;;
closing)
  field() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', 'K', d['steps'], 'ms/step %.5f' % d['ms_per_step'], 'before the closing collective %.5f' % d['ms_per_step_before_closing_barrier'], 'closing collective %.1f us per region' % d['config']['closing_collective_us'], '= %.2f %% of the region' % (100 * d['config']['closing_collective_us'] / (d['ms_per_step'] * d['steps'] * 1e3)), d['config']['exchange'])"; }
  echo "# (ms_per_step - ms_per_step_before_closing_barrier) x K: the time between the LAST rank's device going idle and the closing all-reduce + sync returning, MAX over ranks" | tee $O/closing_cost.txt
  for K in 20 200 20 200; do
    python bench.py --gpus 2 --steps $K --warmup 5 --gather sympush --dist-backend gloo --same-device --prewarm-ms 100 --bodies 65536 --p2p-timeout-ms 5000 2> $O/closing_gloo_$K.err | field "2 ranks on one GPU, gloo, 65536 bodies, sharded symmetric step:" | tee -a $O/closing_cost.txt
  done
  for K in 20 200 20 200; do
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29800 + K % 90)) bench.py --gpus 1 --steps $K --warmup 5 --force-comm --no-cpu-baseline --prewarm-ms 100 2> $O/closing_rccl_$K.err | field "1 rank, RCCL backend (device tensors), 65536 bodies, all-gather behind every step:" | tee -a $O/closing_cost.txt
  done ;;
partial)
  python tools/partial_sweep.py "$@" 2>&1 | tee $O/partial_sweep.txt ;;
partialstats)
  # per-kernel times of a half-active step in its split form (65 536 bodies, 32 768 active; 2000 steps): the one-sided launch over the frozen
  # bodies, the symmetric launch over the active blocks, the reduce launch
  cat > /tmp/partial_steps.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import mapn
n, na = int(sys.argv[1]), int(sys.argv[2])
with mapn.Compute(n, device=0, mass=70000.0 / n) as c:
    c.set_timers(0)
    for _ in range(int(sys.argv[3])):
        c.Simulate(na, c.GetFenceValue())
    c.WaitForGpu()
    assert c.kernel_stats().split_active == na
PY
  n=${1:-65536}; na=${2:-32768}; cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/pstats -- python3 /tmp/partial_steps.py $n $na 2000 > /dev/null 2> $O/pstats.err
  cd $R; f=$(ls -t $(find $O/pstats -name "*kernel_stats.csv") | head -1); cp $f $O/partial_kernel_stats.csv; head -6 $f; rm -rf $O/pstats ;;
rankfail)
  show() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); c=d['config']; print('  ->', 'exchange', c['exchange'], '| trial us/step', {k: round(v, 1) for k, v in c['exchange_trial_us_per_step'].items()}, '| p2p_failure:', c['p2p_failure'], '| fallback_after_failure:', c['fallback_after_failure'], '| replicas bit-identical after the run:', c['replicas_bit_identical_after_run'], '| valid:', c['valid'], '| ms/step %.4f' % d['ms_per_step'])"; }
  : > $O/rank_failure_8_ranks.txt
  for inj in --test-inject-trial-failure --test-inject-push-failure; do
    [ $inj = --test-inject-trial-failure ] && G=p2pall || G=sympush        # (b): the pushed form asked for by name, so that it is the one the timed region runs
    echo "== bench.py --gpus 8 --gather $G $inj (8 processes sharing one GPU, gloo rendezvous, 65536 bodies)" | tee -a $O/rank_failure_8_ranks.txt
    timeout 600 python bench.py --gpus 8 --steps 40 --warmup 5 --gather $G --dist-backend gloo --same-device --no-survey-leg --prewarm-ms 20 --bodies 65536 \
        --p2p-timeout-ms 5000 --xcd off $inj 2> $O/rankfail$inj.err | show | tee -a $O/rank_failure_8_ranks.txt
    grep "^\[bench\]" $O/rankfail$inj.err | cut -c1-400 | sed 's/^/  /' | tee -a $O/rank_failure_8_ranks.txt
  done ;;
power)
  bash tools/power_probe.sh > $O/power_probe.txt 2>&1; cat $O/power_probe.txt ;;
bench)
  python bench.py "$@" > $O/bench.json 2> $O/bench.err; line "bench $*" < $O/bench.json ;;
stats)
  k=${1:-sym}; cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$k -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-partial-leg --kernel $k > $O/bench_profiled_$k.json 2> $O/bench_profiled_$k.err
  cd $R; f=$(ls -t $(find $O/stats_$k -name "*kernel_stats.csv") | head -1); cp $f $O/kernel_stats_$k.csv; head -5 $f; line "profiled $k" < $O/bench_profiled_$k.json
  rm -rf $O/stats_$k ;;     # (the raw trace: gpurun merges at most 64 MiB back)
pmc)
  k=${1:-sym}; tag=${2:-r05_$k}; cd /tmp
  P() { d=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$k/$d -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-partial-leg --kernel $k > /dev/null 2> $O/pmc_$k.$d.err; }   # (--no-partial-leg: that leg launches the SAME force kernel over fewer blocks and would be averaged into its per-launch figures)
  P fetch FETCH_SIZE; P write WRITE_SIZE
  P sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAVES
  P grbm GRBM_GUI_ACTIVE GRBM_COUNT
  cd $R; python tools/pmc_summary.py $O/pmc_$k $tag > $O/pmc_summary_$k.txt 2>&1; tail -30 $O/pmc_summary_$k.txt
  cp profiles/${tag}_pmc_summary.txt profiles/${tag}_pmc_summary.json $O/; rm -rf $O/pmc_$k ;;     # (summaries travel back in gpurun_out; the raw counter files stay on the box)
issue)
  t=${1:-.}; cd /tmp
  P() { d=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$d -- python3 $R/$t/bench.py --steps 12 --warmup 3 --no-cpu-baseline > /dev/null 2> $O/$d.err; }
  P w1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
  P w2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE
  P w3 SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_WAVES SQ_CYCLES
  cd $R; python - "$O" <<'PY' | tee $O/issue.txt
import csv, glob, collections, sys
O = sys.argv[1]
for d in ("w1", "w2", "w3"):
    acc = collections.defaultdict(float); cnt = collections.Counter(); dur = []
    for f in glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "force_sym" not in r["Kernel_Name"]: continue
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for c in sorted(acc): print("%-24s %16.1f per launch" % (c, acc[c] / cnt[c]))
    if dur: print("  (%s: force_sym_kernel %.1f us under the counters)" % (d, sum(dur) / len(dur) / 1e3))
PY
  ;;
ab)
  t=${1:?other tree}
  for rep in 1 2; do
    (cd $t && python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null) | line "$t" | tee -a $O/ab.txt
    python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | line "tree" | tee -a $O/ab.txt
    (cd $t && python tools/shard_sym_loopback.py 65536 300 2>&1 | grep "world 8  sym" | sed "s|^|$t |") | tee -a $O/ab.txt
    python tools/shard_sym_loopback.py 65536 300 2>&1 | grep "world 8  sym" | sed "s|^|tree |" | tee -a $O/ab.txt
  done ;;
timeline)
  python tools/shard_timeline.py "$@" 2>&1 | tee $O/timeline_$(echo "$*" | tr ' ' '_').txt ;;
shardstep)
  n=${1:-65536}; world=${2:-8}; tree=${3:-}; [ -n "$tree" ] && { O=$O/$(basename $tree); mkdir -p $O; export MAPN_TREE=$R/$tree; }   # [TREE]: another checkout, for a same-box A/B
  rm -rf $O/trace_*; cd /tmp
  cat > /tmp/loopstep.py <<'PY'
import os, sys
os.environ["MAPN_TEST_HOOKS"] = "1"; os.environ["MAPN_P2P_LOOPBACK"] = "1"
sys.path.insert(0, os.environ.get("MAPN_TREE") or os.environ["GRAFT_REPO_ROOT"])
import mapn
n, world, algo, rank = (int(x) for x in sys.argv[1:5])
with mapn.Compute(n, device=0, mass=70000.0 / n, rank=rank, world_size=world) as c:
    blob = c.p2p_export(); c.p2p_import([blob] * world); c.set_gather_algorithm(algo); c.set_timers(0)
    for _ in range(2000):                  # 0.2 s: the clock has settled by the second half, which is what is analysed
        c.Simulate(n, c.GetFenceValue())
    c.WaitForGpu()
PY
  for algo in 5 4 2; do for rank in 0 $((world / 2)); do
    rocprofv3 --kernel-trace --output-format csv -d $O/trace_${algo}_$rank -- python3 /tmp/loopstep.py $n $world $algo $rank > /dev/null 2> $O/trace_${algo}_$rank.err
  done; done
  # the forms that need NO mapped peer memory (hipIpc unavailable): rank 0 on a ONE-rank RCCL communicator (MAPN_COMM_LOOPBACK): the
  # launches and the collective calls are real, the wire is not.  6 = sharded symmetric step over RCCL, 0 = one-sided + ncclAllGather, 10 = 0 + overlap
  cat > /tmp/loopstep_comm.py <<'PY'
import os, sys
os.environ["MAPN_TEST_HOOKS"] = "1"; os.environ["MAPN_COMM_LOOPBACK"] = "1"
sys.path.insert(0, os.environ.get("MAPN_TREE") or os.environ["GRAFT_REPO_ROOT"])
import mapn
n, world, algo = (int(x) for x in sys.argv[1:4])
with mapn.Compute(n, device=0, mass=70000.0 / n, rank=0, world_size=world) as c:
    c.comm_init(mapn.Compute.comm_unique_id()); c.set_gather_algorithm(algo % 10); c.set_shard_overlap(algo >= 10); c.set_timers(0)
    for _ in range(2000):
        c.Simulate(n, c.GetFenceValue())
    c.WaitForGpu()
PY
  for algo in 6 0 10; do
    rocprofv3 --kernel-trace --output-format csv -d $O/trace_${algo}_0 -- python3 /tmp/loopstep_comm.py $n $world $algo > /dev/null 2> $O/trace_${algo}_0.err
  done
  cd $R; python - "$O" $n $world <<'PY' | tee $O/shard_step_timeline.txt
import csv, glob, collections, sys
O, n, world = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
print(f"# kernel trace (rocprofv3 --kernel-trace) of one rank's step at {n} / {world} in loopback (every peer mapped to the rank itself), second half of 2000 steps:")
print("# MEDIAN (mean) durations and gaps; step period = median distance between two force launches' starts.  Algorithm 5: the exchange launch pushes the new positions; 4: it pulls them; 2: one-sided kernel, separate pull launch (skipped in loopback)")
names = {5: "5 (symmetric, positions pushed)", 4: "4 (symmetric, positions pulled)", 2: "2 (one-sided kernel + pull)", 6: "6 (symmetric over RCCL alone; 1-rank communicator)",
         0: "0 (one-sided kernel + ncclAllGather; 1-rank communicator)", 10: "0 + overlap (own-segment launch || all-gather; 1-rank communicator)"}
for algo in (5, 4, 2, 6, 0, 10):
    for rank in (0, world // 2):
        f = glob.glob(f"{O}/trace_{algo}_{rank}/**/*kernel_trace.csv", recursive=True)
        if not f: continue
        rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
        rows = rows[len(rows) // 2:]
        dur = collections.defaultdict(list); gap = collections.defaultdict(list)
        for a, b in zip(rows, rows[1:]):
            k = a["Kernel_Name"].split("(")[0][-40:]
            dur[k].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
            gap[k + " -> next"].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
        print(f"== gather algorithm {names[algo]}, rank {rank}")
        med = lambda v: sorted(v)[len(v) // 2]
        for k, v in dur.items(): print("  %-48s %7.2f us  (mean %.2f, x%d)" % (k, med(v) / 1e3, sum(v) / len(v) / 1e3, len(v)))
        for k, v in gap.items(): print("  gap %-44s %7.2f us  (mean %.2f)" % (k, med(v) / 1e3, sum(v) / len(v) / 1e3))
        top = collections.Counter(r["Kernel_Name"] for r in rows if "force" in r["Kernel_Name"]).most_common(1)[0][0]
        starts = [int(r["Start_Timestamp"]) for r in rows if r["Kernel_Name"] == top]
        per = 2 if algo == 10 and sum("force" in k for k in dur) == 1 else 1      # (overlap structure: two launches of one kernel per step)
        starts = starts[::per]
        d = [b - a for a, b in zip(starts, starts[1:])]
        print("  step period %.2f us  (mean %.2f)" % (med(d) / 1e3, sum(d) / len(d) / 1e3))
PY
  ;;
shardpartial)
  n=${1:-65536}; world=${2:-8}; rm -rf $O/trace_*; cd /tmp
  cat > /tmp/loopstep_partial.py <<'PY'
import os, sys
os.environ["MAPN_TEST_HOOKS"] = "1"; os.environ["MAPN_P2P_LOOPBACK"] = "1"
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import mapn
n, world, algo, rank, na = (int(x) for x in sys.argv[1:6])
with mapn.Compute(n, device=0, mass=70000.0 / n, rank=rank, world_size=world) as c:
    blob = c.p2p_export(); c.p2p_import([blob] * world); c.set_gather_algorithm(algo); c.set_timers(0)
    for _ in range(2000):
        c.Simulate(na, c.GetFenceValue())
    c.WaitForGpu()
    print("split_active", c.kernel_stats().split_active, c.kernel_stats().kernel_name.decode())
PY
  for na in $((n / 2)) $((n * 5 / 8)); do for cfg in "5 0" "5 $((world - 1))" "2 0"; do
    set -- $cfg; algo=$1; rank=$2
    rocprofv3 --kernel-trace --output-format csv -d $O/trace_${na}_${algo}_$rank -- python3 /tmp/loopstep_partial.py $n $world $algo $rank $na > $O/trace_${na}_${algo}_$rank.out 2> $O/trace_${na}_${algo}_$rank.err
  done; done
  cd $R; python - "$O" $n $world <<'PY' | tee $O/shard_partial_timeline.txt
import csv, glob, collections, sys
O, n, world = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
print(f"# PARTIALLY ACTIVE steps of the {n}-body job over {world} ranks, one rank at a time in loopback (every peer mapped to the rank itself): rocprofv3 --kernel-trace,")
print("# second half of 2000 steps, MEDIAN (mean) durations and gaps.  Algorithm 5 = the split form (round 6): an ACTIVE rank runs its blocks of the active ring under the symmetric")
print("# kernel, a FROZEN rank the one-sided launch over active x (its frozen bodies); both run the exchange launch.  Algorithm 2 = rounds 1 - 5's step: the one-sided kernel over")
print("# (the rank's active bodies) x N (the pull launch is skipped in loopback).  The JOB's step is its slowest rank's.")
for na in (n // 2, n * 5 // 8):
    periods = {}
    for algo, rank, what in ((5, 0, "split form, rank 0 (all its bodies active)"), (5, world - 1, f"split form, rank {world - 1} (all its bodies frozen)"), (2, 0, "algorithm 2, rank 0")):
        f = glob.glob(f"{O}/trace_{na}_{algo}_{rank}/**/*kernel_trace.csv", recursive=True)
        if not f: continue
        rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
        rows = rows[len(rows) // 2:]
        dur = collections.defaultdict(list)
        for a in rows: dur[a["Kernel_Name"].split("(")[0][-40:]].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
        med = lambda v: sorted(v)[len(v) // 2]
        print(f"== num_active {na}: {what}")
        for k, v in dur.items(): print("  %-48s %7.2f us  (mean %.2f, x%d)" % (k, med(v) / 1e3, sum(v) / len(v) / 1e3, len(v)))
        top = collections.Counter(r["Kernel_Name"] for r in rows if "force" in r["Kernel_Name"]).most_common(1)[0][0]
        starts = [int(r["Start_Timestamp"]) for r in rows if r["Kernel_Name"] == top]
        d = [b - a for a, b in zip(starts, starts[1:])]
        periods[(algo, rank)] = med(d) / 1e3
        print("  step period %.2f us  (mean %.2f)" % (med(d) / 1e3, sum(d) / len(d) / 1e3))
    if (5, 0) in periods and (5, world - 1) in periods and (2, 0) in periods:
        job = max(periods[(5, 0)], periods[(5, world - 1)])
        print(f"## num_active {na}: the job's step in the split form = max({periods[(5, 0)]:.2f}, {periods[(5, world - 1)]:.2f}) = {job:.2f} us against {periods[(2, 0)]:.2f} us for algorithm 2: {periods[(2, 0)] / job:.3f} x")
PY
  ;;
loopback)
  python tools/shard_sym_loopback.py ${1:-65536} ${2:-400} 2>&1 | tee $O/loopback_${1:-65536}.txt ;;
sizes)
  for n in 65536 100000 262144 1048576 4194304; do
    st=$((n > 1000000 ? 4 : n > 200000 ? 30 : 200)); wu=$((n > 1000000 ? 1 : 5))
    for k in sym sgpr; do python bench.py --bodies $n --steps $st --warmup $wu --prewarm-ms $((n > 1000000 ? 0 : 400)) --no-cpu-baseline --kernel $k 2>/dev/null | line "$n $k" | tee -a $O/sizes.txt; done
  done ;;
ubench)
  [ tools/ubench -nt tools/ubench.hip ] || hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench
  cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $R/tools/ubench --ab 20000 > $O/ubench_ab.jsonl 2> $O/ubench_ab.err
  cd $R; f=$(ls -t $(find $O/stats -name "*kernel_stats.csv") | head -1); cp $f $O/ubench_ab_kernel_stats.csv; head -30 $f
  python -m pytest tests/test_gpu_mfma_ab.py -m gpu -q -s 2>&1 | tee $O/pytest_mfma_ab.txt | tail -25
  cp gpurun_out/ubench_ab_test_report.txt $O/ 2>/dev/null ;;
parity1000)
  python tests/parity_report.py --bodies 65536 --steps 1,10,100,1000 --f64-max-steps 100 --out $O/parity_1000_65536.json > $O/parity_1000.txt 2>&1; tail -30 $O/parity_1000.txt ;;
soak)
  mode=${1:-sympush}; mkdir -p /tmp/soak
  # (several processes share ONE GPU here, each with its own hardware queue: the library sizes its waiting launches for that --
  #  mapn_p2p_import counts the ranks on this device -- or eight of them fill the device waiting for each other: DESIGN 5)
  for cfg in "2 8192 1500" "4 8192 1500" "8 8192 1500" "8 16384 600" "2 32768 600" "4 32768 600" "2 65536 400" "8 65536 300"; do
    set -- $cfg; Wd=$1; N=$2; S=$3
    for m in p2p $mode; do
      rm -rf /tmp/soak/$m; mkdir -p /tmp/soak/$m; pids=""
      for r in $(seq 0 $((Wd - 1))); do python tests/shard_gpu_worker.py $r $Wd $((29850 + Wd)) $N $S /tmp/soak/$m $m > /tmp/soak/$m/log_$r.txt 2>&1 & pids="$pids $!"; done
      t0=$(date +%s); ok=1; for p in $pids; do wait $p || ok=0; done
      [ $ok = 1 ] && echo "world=$Wd n=$N steps=$S mode=$m: done in $(( $(date +%s) - t0 )) s" | tee -a $O/soak_$mode.txt
      [ $ok = 1 ] || { echo "world=$Wd n=$N steps=$S mode=$m FAILED" | tee -a $O/soak_$mode.txt; for f in /tmp/soak/$m/log_*.txt; do tail -n 2 $f; done; }
    done
    python - $Wd $N $S $mode <<'PY' | tee -a $O/soak_$mode.txt
import sys, numpy as np
Wd, N, S, mode = sys.argv[1:5]
a = np.load("/tmp/soak/p2p/gpu_sharded.npz"); b = np.load(f"/tmp/soak/{mode}/gpu_sharded.npz")
same = all(np.array_equal(a[k], b[k]) for k in ("pos", "vel", "other"))
d = np.linalg.norm(a["pos"][:, :3].astype(np.float64) - b["pos"][:, :3], axis=1) / np.maximum(np.linalg.norm(a["pos"][:, :3].astype(np.float64), axis=1), 1e-30)
print(f"world={Wd} n={N} steps={S}  {mode} vs p2p: bitwise equal {same}, relative position difference max {d.max():.2e} median {np.median(d):.2e}, finite {bool(np.isfinite(b['pos']).all())}")
PY
  done ;;
slider)
  python tools/slider_drag.py 2>&1 | tee $O/slider_drag.txt ;;
soakslider)
  mkdir -p /tmp/soak
  for cfg in "8 65536 700" "4 65536 500" "2 32768 500" "8 16384 1400"; do
    set -- $cfg; Wd=$1; N=$2; S=$3
    export MAPN_WORKER_SLIDER="$N,$((N/2)),$((N/2)),$((5*N/8)),$((N/2+1000)),1500,0,$((3*N/8)),$((3*N/4)),$((7*N/8)),$((N/4+64)),$((N/2)),$N,$((5*N/8)),$((N/2+64)),$((N-64)),$N"
    for m in sym sympush; do
      rm -rf /tmp/soak/$m; mkdir -p /tmp/soak/$m; pids=""
      for r in $(seq 0 $((Wd - 1))); do python tests/shard_gpu_worker.py $r $Wd $((29850 + Wd)) $N $S /tmp/soak/$m $m $N > /tmp/soak/$m/log_$r.txt 2>&1 & pids="$pids $!"; done
      t0=$(date +%s); ok=1; for p in $pids; do wait $p || ok=0; done
      [ $ok = 1 ] && echo "world=$Wd n=$N steps=$S mode=$m: done in $(( $(date +%s) - t0 )) s" | tee -a $O/slider_soak.txt
      [ $ok = 1 ] || { echo "world=$Wd n=$N steps=$S mode=$m FAILED" | tee -a $O/slider_soak.txt; for f in /tmp/soak/$m/log_*.txt; do tail -n 3 $f; done; }
    done
    python - $Wd $N $S <<'PY' | tee -a $O/slider_soak.txt
import sys, numpy as np
Wd, N, S = sys.argv[1:4]
a = np.load("/tmp/soak/sym/gpu_sharded.npz"); b = np.load("/tmp/soak/sympush/gpu_sharded.npz")
same = all(np.array_equal(a[k], b[k]) for k in ("pos", "vel", "other"))
print(f"world={Wd} n={N} steps={S}, slider sequence of 17 counts cycled: pushed vs pulled bitwise equal {same}, finite {bool(np.isfinite(b['pos']).all())}")
PY
  done ;;
*) echo "unknown: $W"; exit 2 ;;
esac
