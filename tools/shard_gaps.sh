#!/bin/bash
# kernel timeline of rank 0's sharded symmetric step at 65 536 / 8 in loopback: durations and gaps
R=$PWD; O=$R/gpurun_out/r02t; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
cat > /tmp/loop8.py <<'PY'
import os, sys
os.environ["MAPN_P2P_LOOPBACK"] = "1"
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import mapn
n, world = 65536, 8
with mapn.Compute(n, device=0, mass=70000.0 / n, rank=0, world_size=world) as c:
    blob = c.p2p_export(); c.p2p_import([blob] * world); c.set_gather_algorithm(int(sys.argv[1])); c.set_timers(0)
    for _ in range(300):
        c.Simulate(n, c.GetFenceValue())
    c.WaitForGpu()
PY
for algo in 4 2; do rocprofv3 --kernel-trace --output-format csv -d $O/trace_$algo -- python3 /tmp/loop8.py $algo > /dev/null 2> $O/trace_$algo.err; done
cd $R
python - <<'PY'
import csv, glob, collections
for algo in (4, 2):
    f = glob.glob(f"gpurun_out/r02t/trace_{algo}/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[len(rows)//2:]                      # steady state
    dur = collections.defaultdict(list); gap = collections.defaultdict(list)
    for a, b in zip(rows, rows[1:]):
        k = a["Kernel_Name"].split("(")[0][-40:]
        dur[k].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
        gap[k + " -> next"].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
    print(f"== gather algorithm {algo}")
    for k, v in dur.items(): print("  %-48s %7.2f us  (x%d)" % (k, sum(v)/len(v)/1e3, len(v)))
    for k, v in gap.items(): print("  gap %-44s %7.2f us" % (k, sum(v)/len(v)/1e3))
    t = (int(rows[-1]["Start_Timestamp"]) - int(rows[0]["Start_Timestamp"]))
    per = collections.Counter(r["Kernel_Name"] for r in rows).most_common(1)[0][1]
    print("  step period %.2f us" % (t / per / 1e3))
PY
