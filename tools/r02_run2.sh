#!/bin/bash
# round 2, second GPU pass: full -m gpu suite (incl. the 1000-step parity), shard-size sweeps ticket vs rows
R=$PWD; O=$R/gpurun_out/r02b; mkdir -p $O
python -m pytest tests -m "gpu and not slow" -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -15 $O/pytest_gpu.txt
python -m pytest tests/test_parity_1000.py -m gpu -q -s > $O/pytest_parity1000.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_parity1000.txt
tail -30 $O/pytest_parity1000.txt
# one rank's share of the 8-way 65 536-body job: rows vs ticket, external gather (no exchange cost)
for fused in 0 1; do
  MAPN_EPILOGUE=$([ $fused = 0 ] && echo rows || echo ticket) python tools/sweep.py --bodies 65536 --world 8 --auto --steps 400 --timer-interval 0 > $O/shard8192_fused$fused.txt 2>&1
  tail -2 $O/shard8192_fused$fused.txt
done
