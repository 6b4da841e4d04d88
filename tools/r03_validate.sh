#!/bin/bash
# round 3, step 2: the rewritten symmetric kernels (step-exact deal from host-built plan tables, windows, exchange + pull in one
# launch): parity suites first, then the timelines and the kernel trace of the loopback step.
R=$PWD; O=$R/gpurun_out/r03b; rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_sym.py -x -q > $O/pytest_sym.txt 2>&1; tail -5 $O/pytest_sym.txt
python -m pytest tests/test_shard_gpu_multiproc.py -x -q > $O/pytest_shard.txt 2>&1; tail -5 $O/pytest_shard.txt
python tools/shard_timeline.py 65536 8 0 4 > $O/timeline_rank0.txt 2>&1
python tools/shard_timeline.py 65536 8 4 4 > $O/timeline_rank4.txt 2>&1
MAPN_SYM_SHARD_PULL=0 python tools/shard_timeline.py 65536 8 0 4 > $O/timeline_rank0_nopull.txt 2>&1
python tools/shard_timeline.py 65536 1 0 0 > $O/timeline_unsharded.txt 2>&1
head -3 $O/timeline_rank0.txt $O/timeline_rank4.txt $O/timeline_rank0_nopull.txt $O/timeline_unsharded.txt
cd /tmp; export TMPDIR=/tmp
cat > /tmp/loop8.py <<'PY'
import os, sys
os.environ["MAPN_P2P_LOOPBACK"] = "1"
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import mapn
n, world, rank = 65536, 8, int(sys.argv[2])
with mapn.Compute(n, device=0, mass=70000.0 / n, rank=rank, world_size=world) as c:
    blob = c.p2p_export(); c.p2p_import([blob] * world); c.set_gather_algorithm(int(sys.argv[1])); c.set_timers(0)
    for _ in range(300):
        c.Simulate(n, c.GetFenceValue())
    c.WaitForGpu()
PY
for rank in 0 4; do rocprofv3 --kernel-trace --output-format csv -d $O/trace_$rank -- python3 /tmp/loop8.py 4 $rank > /dev/null 2> $O/trace_$rank.err; done
cd $R
python - <<'PY'
import csv, glob, collections
for rank in (0, 4):
    f = glob.glob(f"gpurun_out/r03b/trace_{rank}/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[len(rows)//2:]                      # steady state
    dur = collections.defaultdict(list); gap = collections.defaultdict(list)
    for a, b in zip(rows, rows[1:]):
        k = a["Kernel_Name"].split("(")[0][-40:]
        dur[k].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
        gap[k + " -> next"].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
    print(f"== rank {rank}, gather algorithm 4, loopback")
    for k, v in dur.items(): print("  %-48s %7.2f us  (x%d)" % (k, sum(v)/len(v)/1e3, len(v)))
    for k, v in gap.items(): print("  gap %-44s %7.2f us" % (k, sum(v)/len(v)/1e3))
    t = (int(rows[-1]["Start_Timestamp"]) - int(rows[0]["Start_Timestamp"]))
    per = collections.Counter(r["Kernel_Name"] for r in rows).most_common(1)[0][1]
    print("  step period %.2f us" % (t / per / 1e3))
PY
