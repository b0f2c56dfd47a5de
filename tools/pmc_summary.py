#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output: per kernel, the average of every counter over its launches; and, under "<counter>_big" /
"launches_big", the same over the kernel's LARGE launches only (one criterion per kernel for every counter: the launch's grid size, else its duration, >= 0.6 x
the kernel's largest): k_col_strided<N1, 1> runs
both as a four-field launch per RK stage and as single-field launches of set/get, and bench.py prices the former.
usage: pmc_summary.py <dir-with-*_counter_collection.csv> [more dirs...]  -> JSON on stdout"""
import csv, glob, json, os, re, sys
from collections import defaultdict

def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)

acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
rows = defaultdict(lambda: defaultdict(list))              # kernel -> counter -> [(grid size or None, value), ...]
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k, c = short(row["Kernel_Name"]), row["Counter_Name"]
                a = acc[k][c]
                a[0] += float(row["Counter_Value"]); a[1] += 1
                g, t0, t1 = row.get("Grid_Size"), row.get("Start_Timestamp"), row.get("End_Timestamp")
                dur = (float(t1) - float(t0)) if t0 not in (None, "") and t1 not in (None, "") else None
                rows[k][c].append((float(g) if g not in (None, "") else None, dur, float(row["Counter_Value"])))
out = {k: {c: v[0] / v[1] for c, v in cs.items()} | {"launches": max(v[1] for v in cs.values())} for k, cs in acc.items()}
for k, cs in rows.items():
    # The LARGE launches of a kernel are chosen by ONE criterion for every counter, taken from what every pass records about a launch
    # itself: its grid size (>= 0.6 x the kernel's largest grid) or, where the grids do not tell the launches apart (a persistent grid),
    # its duration (>= 0.6 x the longest launch of that pass: a four-field launch of k_col_strided runs four times as long as a one-field
    # one).  The passes are separate processes whose launch COUNTS can differ (the pitch probe of fb_create is time-based), so neither
    # dispatch ids nor launch order match across them.  Only as a last resort does the counter's own value decide (>= 0.6 x
    # its maximum), as in round 3.
    grids = [g for v in cs.values() for g, _, _ in v if g is not None]
    by_grid = bool(grids) and min(grids) < 0.6 * max(grids)
    by_time = not by_grid and all(d is not None for v in cs.values() for _, d, _ in v) and \
        all(min(d for _, d, _ in v) < 0.6 * max(d for _, d, _ in v) for v in cs.values())
    out[k]["big_selected_by"] = "Grid_Size" if by_grid else ("duration" if by_time else "counter value")
    counts = []
    for c, v in cs.items():
        if by_grid:
            sel = [x for g, _, x in v if g is not None and g >= 0.6 * max(grids)]
        elif by_time:
            top = max(d for _, d, _ in v)
            sel = [x for _, d, x in v if d >= 0.6 * top]
        else:
            top = max(x for _, _, x in v)
            sel = [x for _, _, x in v if x >= 0.6 * top]
        if sel:
            out[k][c + "_big"] = sum(sel) / len(sel)
            counts.append(len(sel))
    out[k]["launches_big"] = max(counts) if counts else 0
    if counts and min(counts) != max(counts):
        out[k]["launches_big_per_counter"] = counts           # the passes saw different numbers of large launches (reported, not hidden)
json.dump(out, sys.stdout, indent=1, sort_keys=True)
