#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output: per kernel, the average of every counter over its launches; and, under "<counter>_big" /
"launches_big", the same over the kernel's LARGE launches only (counter >= 0.6 x the kernel's maximum): k_col_strided<N1, 1> runs
both as a four-field launch per RK stage and as single-field launches of set/get, and bench.py prices the former.
usage: pmc_summary.py <dir-with-*_counter_collection.csv> [more dirs...]  -> JSON on stdout"""
import csv, glob, json, os, re, sys
from collections import defaultdict

def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)

acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
vals = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                a = acc[short(row["Kernel_Name"])][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"]); a[1] += 1
                vals[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: v[0] / v[1] for c, v in cs.items()} | {"launches": max(v[1] for v in cs.values())} for k, cs in acc.items()}
for k, cs in vals.items():
    for c, v in cs.items():
        big = [x for x in v if x >= 0.6 * max(v)]
        out[k][c + "_big"] = sum(big) / len(big)
        out[k]["launches_big"] = len(big)
json.dump(out, sys.stdout, indent=1, sort_keys=True)
