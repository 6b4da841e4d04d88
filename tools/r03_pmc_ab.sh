#!/bin/bash
# same-box issue counters of force_sym_kernel: round 2's kernel (ab/old) against the working tree (PMC passes only)
R=$PWD; O=$R/gpurun_out/r03d; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
P() { t=$1; d=$2; shift 2; mkdir -p $O/$t; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$t/$d -- python3 $R/$t/bench.py --steps 12 --warmup 3 --no-cpu-baseline > /dev/null 2> $O/$t/$d.err; }
ln -sfn $R $R/ab/new
for t in ab/old ab/new; do
  P $t w1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
  P $t w2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE
  P $t w3 SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_WAVES SQ_CYCLES
done
cd $R
python - <<'PY'
import csv,glob,collections,os
O="gpurun_out/r03d"
res={}
for t in ("ab/old","ab/new"):
    for d in ("w1","w2","w3"):
        acc=collections.defaultdict(float); cnt=collections.Counter(); dur=[]
        for f in glob.glob(f"{O}/{t}/{d}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "force_sym" not in r["Kernel_Name"]: continue
                acc[r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[r["Counter_Name"]]+=1
                dur.append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
        for c,x in acc.items(): res[(t,c)]=x/cnt[c]
        if dur: res[(t,"dur_us_"+d)]=sum(dur)/len(dur)/1e3
names=sorted({c for (_,c) in res})
print("%-26s %16s %16s %8s"%("counter","old","new","new/old"))
for c in names:
    a,b=res.get(("ab/old",c)),res.get(("ab/new",c))
    if a is None or b is None: print(c,a,b); continue
    print("%-26s %16.1f %16.1f %8.3f"%(c,a,b,b/a if a else 0))
PY
