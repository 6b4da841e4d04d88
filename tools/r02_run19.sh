#!/bin/bash
# residency vs VALU activity of force_sym_kernel for several workgroup counts and sizes (one --pmc pass each)
R=$PWD; O=$R/gpurun_out/r02u; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
run() { tag=$1; shift; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/$tag -- python3 $R/bench.py --no-cpu-baseline "$@" > /dev/null 2> $O/$tag.err; }
MAPN_SYM_PLAN=4,8 run p8 --steps 12 --warmup 3
MAPN_SYM_PLAN=4,16 run p16 --steps 12 --warmup 3
MAPN_SYM_PLAN=4,32 run p32 --steps 12 --warmup 3
MAPN_SYM_PLAN=4,64 run p64 --steps 12 --warmup 3
run n262144 --bodies 262144 --steps 6 --warmup 2
run n1048576 --bodies 1048576 --steps 3 --warmup 1
cd $R
python - <<'PY'
import csv,glob,collections
O="gpurun_out/r02u"
for d in ("p8","p16","p32","p64","n262144","n1048576"):
    acc=collections.defaultdict(float); cnt=collections.Counter()
    for f in glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "force_sym" not in r["Kernel_Name"]: continue
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[r["Counter_Name"]]+=1
    v={k:acc[k]/cnt[k] for k in acc}
    if not v: print(d,"no data"); continue
    cyc=v["SQ_CYCLES"]/32.0
    slots=2*1024*cyc/4.0
    print("%-9s kernel %.3f Mcycles  residency %.3f  VALU/resident-pair %.3f  VALU busy %.3f  wait_inst %.3f wait_any %.3f  insts %.1fM" % (
        d, cyc/1e6, v["SQ_WAVE_CYCLES"]/slots, v["SQ_ACTIVE_INST_VALU"]/(v["SQ_WAVE_CYCLES"]/2), v["SQ_ACTIVE_INST_VALU"]/(slots/2), v["SQ_WAIT_INST_ANY"]/v["SQ_WAVE_CYCLES"], v["SQ_WAIT_ANY"]/v["SQ_WAVE_CYCLES"], v["SQ_INSTS_VALU"]/1e6))
PY
