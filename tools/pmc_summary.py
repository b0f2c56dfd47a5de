#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output: per kernel, the average of every counter over its launches.
usage: pmc_summary.py <dir-with-*_counter_collection.csv> [more dirs...]  -> JSON on stdout"""
import csv, glob, json, os, re, sys
from collections import defaultdict

def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)

acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                a = acc[short(row["Kernel_Name"])][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"]); a[1] += 1
out = {k: {c: v[0] / v[1] for c, v in cs.items()} | {"launches": max(v[1] for v in cs.values())} for k, cs in acc.items()}
json.dump(out, sys.stdout, indent=1, sort_keys=True)
