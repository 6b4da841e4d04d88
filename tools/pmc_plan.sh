#!/bin/bash
# PMC SQ counters for a list of force plans at 65 536 bodies (run on the GPU box).
R=$PWD; mkdir -p gpurun_out/pmcplan; cd /tmp; export TMPDIR=/tmp
for plan in "$@"; do
  tag=${plan//,/_}
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmcplan/$tag -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --plan $plan > /dev/null 2>&1
done
cd $R
python3 - <<'PY'
import csv,glob,collections,os
for d in sorted(glob.glob("gpurun_out/pmcplan/*")):
    f=glob.glob(d+"/**/*counter_collection.csv",recursive=True)
    if not f: continue
    agg=collections.defaultdict(list); dur=[]
    for r in csv.DictReader(open(f[0])):
        if "force" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"])); dur.append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
    print(os.path.basename(d), "dur_us=%.1f"%(sum(dur)/len(dur)/1e3), " ".join(f"{k}={sum(v)/len(v):.3e}" for k,v in sorted(agg.items())))
PY
