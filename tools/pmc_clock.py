#!/usr/bin/env python3
"""Development aid: per-kernel mean duration and counter values from a rocprofv3 --pmc ... --kernel-trace output directory;
with GRBM_GUI_ACTIVE (or SQ_BUSY_CYCLES) the ratio cycles / duration is the shader clock during the kernel.
Usage: python tools/pmc_clock.py <dir> [kernel-name-substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "k_indirect"
dur = defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        if sub in r["Kernel_Name"]:
            dur[r["Kernel_Name"][:70]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
cnt = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        if sub in r["Kernel_Name"]:
            cnt[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in dur.items():
    v = v[len(v) // 4:]
    m = sum(v) / len(v)
    line = "%-72s n=%4d  dur %8.1f ns" % (k, len(v), m)
    for c, vals in cnt.get(k, {}).items():
        vals = vals[len(vals) // 4:]
        cm = sum(vals) / len(vals)
        line += "  %s=%.4g" % (c, cm)
        if c in ("GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "GRBM_COUNT"):
            line += " (%.3f GHz)" % (cm / m)
    print(line)
