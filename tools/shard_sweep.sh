mkdir -p gpurun_out; export MAPN_COMM_LOOPBACK=1
run() { python tools/sweep.py --world 8 --steps 300 --auto --kernels sgpr --ks 2 --waves 8 --sbs 1 --timer-interval 0 2>/dev/null | grep "step" | sed "s/^/own=$MAPN_OWN_PLAN rem=$MAPN_REM_PLAN  /" | cut -c1-110; }
for own in 2,16,1 2,16,2 2,16,4 2,8,2 2,8,4 2,4,4 4,8,2 4,16,1; do for rem in 2,16,8 2,16,4 2,8,8 2,16,16 4,8,8; do MAPN_OWN_PLAN=$own MAPN_REM_PLAN=$rem run; done; done > gpurun_out/shard_plan_sweep.txt
sort -t'p' -k4 gpurun_out/shard_plan_sweep.txt | awk '{print $0}' | sort -k10 -g | head -12
