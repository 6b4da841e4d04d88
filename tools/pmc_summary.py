#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc passes (one counter group per pass, as the MI355X guide requires)
into profiles/<tag>_pmc_summary.{txt,json}.  Usage: tools/pmc_summary.py gpurun_out/pmc r01

gfx950 corrections applied (MI355X_MICROARCH.md, HBM section): FETCH_SIZE is reported in KiB and
counts 64 B per 128-B request, so bytes read = 2 x FETCH_SIZE x 1024; WRITE_SIZE (KiB) is exact
for 16-B-per-lane streaming stores.  Infinity-Cache hits are counted as fetches.
"""
import collections
import csv
import glob
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "mapn" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["_dur_ns_" + os.path.basename(os.path.dirname(os.path.dirname(f)))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {}
lines = []
for k, cs in sorted(agg.items()):
    d = {c: sum(v) / len(v) for c, v in cs.items()}
    d["launches"] = max(len(v) for v in cs.values())
    if "FETCH_SIZE" in d:
        d["hbm_read_bytes_per_launch"] = 2 * d["FETCH_SIZE"] * 1024
    if "WRITE_SIZE" in d:
        d["hbm_write_bytes_per_launch"] = d["WRITE_SIZE"] * 1024
    if "hbm_read_bytes_per_launch" in d and "hbm_write_bytes_per_launch" in d:
        d["hbm_bytes_per_launch"] = d["hbm_read_bytes_per_launch"] + d["hbm_write_bytes_per_launch"]
    # (GRBM_GUI_ACTIVE also counts the front end's work around a kernel: for a launch of a few microseconds "cycles / duration" is
    #  not a clock -- round 3's summary showed 4.27 GHz for a 10 us kernel -- so no clock, and nothing derived from one, below 100 us)
    if "GRBM_GUI_ACTIVE" in d and d.get("_dur_ns_grbm", 0) >= 100e3:
        d["effective_clock_ghz"] = d["GRBM_GUI_ACTIVE"] / 8 / d["_dur_ns_grbm"]
    if "SQ_ACTIVE_INST_VALU" in d and "_dur_ns_sq" in d and "effective_clock_ghz" in d:
        # SQ_ACTIVE_INST_* counts quad-cycles summed over all SIMDs (1024 on MI355X)
        d["valu_busy_fraction"] = 4 * d["SQ_ACTIVE_INST_VALU"] / 1024 / (d["_dur_ns_sq"] * d["effective_clock_ghz"])
    out[k] = d
    lines.append(k)
    for c in sorted(d):
        lines.append(f"    {c:32s} {d[c]:20.3f}")
# the kernel source these counters were taken from: bench.py drops a `traffic` figure whose sha does
# not match the kernels it is running (a committed traffic number goes stale)
import hashlib
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_h = hashlib.sha256()
for _f in ("mapn_kernels.hip", "mapn_sym.hip"):
    _h.update(open(os.path.join(here, "multi-adapter-particles_amd", "csrc", _f), "rb").read())
out["_kernel_source_sha16"] = _h.hexdigest()[:16]
lines.append(f"kernel source sha16 (multi-adapter-particles_amd/csrc/mapn_kernels.hip + mapn_sym.hip): {out['_kernel_source_sha16']}")
os.makedirs("profiles", exist_ok=True)
open(f"profiles/{tag}_pmc_summary.txt", "w").write(__doc__ + "\n" + "\n".join(lines) + "\n")
json.dump(out, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1)
print("\n".join(lines))
