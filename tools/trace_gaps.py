#!/usr/bin/env python3
"""Prints the last N dispatches of a rocprofv3 kernel trace with durations and the idle gap
before each (development tool).  Usage: tools/trace_gaps.py <dir with *_kernel_trace.csv> [N]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-n:]
prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f'{r["Kernel_Name"][:70]:70s} dur={(e - s) / 1e3:8.1f} us  gap_before={gap:6.1f} us  grid={r["Grid_Size_X"]},{r["Grid_Size_Y"]},{r["Grid_Size_Z"]} wg={r["Workgroup_Size_X"]}')
    prev = e
