#!/bin/bash
# where do the symmetric kernel's non-VALU cycles go: SQ wait / issue counters (PMC passes only)
R=$PWD; O=$R/gpurun_out/r02n; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
P() { d=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$d -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline > /dev/null 2> $O/$d.err; }
P w1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
P w2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE
P w3 SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_WAVES SQ_CYCLES
cd $R
python - <<'PY'
import csv,glob,collections,os
O="gpurun_out/r02n"
for d in ("w1","w2","w3"):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for f in glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:40]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
    for k,v in acc.items():
        if "force_sym" not in k: continue
        for c,x in sorted(v.items()): print(d, k, c, "%.1f per launch" % (x/cnt[(k,c)]))
    err=open(f"{O}/{d}.err").read()
    if "rror" in err: print(d, "ERR", err[-600:])
PY
grep -ci "SQ_" $O/counters_list.txt
