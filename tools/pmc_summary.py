#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output: per kernel, the average of every counter over its launches; and, under "<counter>_big" /
"launches_big", the same over the kernel's LARGE launches only (one set per kernel: launches whose FETCH_SIZE is >= 0.6 x the kernel's
maximum, matched across the counter passes by launch order): k_col_strided<N1, 1> runs
both as a four-field launch per RK stage and as single-field launches of set/get, and bench.py prices the former.
usage: pmc_summary.py <dir-with-*_counter_collection.csv> [more dirs...]  -> JSON on stdout"""
import csv, glob, json, os, re, sys
from collections import defaultdict

def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)

acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
vals = defaultdict(lambda: defaultdict(dict))              # kernel -> counter -> {dispatch key: value}
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        order = defaultdict(int)
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k, c = short(row["Kernel_Name"]), row["Counter_Name"]
                a = acc[k][c]
                a[0] += float(row["Counter_Value"]); a[1] += 1
                # counters come from separate passes of the same command: the n-th launch of a kernel in one pass is the n-th in the other
                order[(k, c)] += 1
                vals[k][c][order[(k, c)]] = vals[k][c].get(order[(k, c)], 0.0) + float(row["Counter_Value"])
out = {k: {c: v[0] / v[1] for c, v in cs.items()} | {"launches": max(v[1] for v in cs.values())} for k, cs in acc.items()}
for k, cs in vals.items():
    # ONE set of large launches per kernel, chosen by a reference counter (the bytes read, else the first counter) and keyed by launch
    # order; every counter is averaged over that same set (a per-counter threshold could pick different launches)
    ref = "FETCH_SIZE" if "FETCH_SIZE" in cs else sorted(cs)[0]
    top = max(cs[ref].values())
    big = sorted(i for i, x in cs[ref].items() if x >= 0.6 * top)
    out[k]["launches_big"] = len(big)
    out[k]["big_selected_by"] = ref
    for c, v in cs.items():
        sel = [v[i] for i in big if i in v]
        if sel:
            out[k][c + "_big"] = sum(sel) / len(sel)
json.dump(out, sys.stdout, indent=1, sort_keys=True)
