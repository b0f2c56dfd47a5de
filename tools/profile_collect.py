#!/usr/bin/env python3
"""Copy the summaries of tools/profile_round.sh from gpurun_out/prof_<tag>/ into profiles/<tag>_*.
usage: profile_collect.py <tag> [grid]"""
import json, os, shutil, sys
tag = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_%s_%d" % (tag, n)); dst = os.path.join(root, "profiles")
import subprocess
commit = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], stdout=subprocess.PIPE, text=True).stdout.strip()
dirty = bool(subprocess.run(["git", "-C", root, "status", "--porcelain", "--", "xlab-fftbarotropic_amd", "bench.py"], stdout=subprocess.PIPE, text=True).stdout.strip())
shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(dst, "%s_kernel_stats_%d.csv" % (tag, n)))
line = [l for l in open(os.path.join(src, "bench.json")) if l.startswith("{")][-1]
json.dump(json.loads(line), open(os.path.join(dst, "%s_bench_%d.json" % (tag, n)), "w"), indent=1)
pmc = json.load(open(os.path.join(src, "pmc.json")))
P = 16 * ((n // 2 + 1 + 15) // 16)
C = 8 * n * P                                   # one complex field at the un-tuned pitch
out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 3 --warmup 1 --cpu-steps 0 (%d^2, separate passes)" % n,
       "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE counts 128-B requests as 64 B)",
       "C_bytes": C, "kernels": {},
       "commit": commit + ("+uncommitted changes" if dirty else ""), "note": "sources of the profiled run = this commit (the profile is collected before it is committed; '+uncommitted' = the working tree at collection time)"}
for k, v in pmc.items():
    if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    b = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
    out["kernels"][k] = {"FETCH_SIZE_KB_avg": v["FETCH_SIZE"], "WRITE_SIZE_KB_avg": v["WRITE_SIZE"], "launches": v["launches"],
                         "hbm_bytes_per_launch_corrected": b, "in_units_of_C": round(b / C, 2)}
    if "FETCH_SIZE_big" in v and "WRITE_SIZE_big" in v:        # the kernel's large launches only (four-field k_col_strided<N1, 1>)
        bb = (2 * v["FETCH_SIZE_big"] + v["WRITE_SIZE_big"]) * 1024
        out["kernels"][k].update({"hbm_bytes_per_big_launch_corrected": bb, "big_in_units_of_C": round(bb / C, 2), "launches_big": v["launches_big"]})
json.dump(out, open(os.path.join(dst, "%s_pmc_traffic_%d.json" % (tag, n)), "w"), indent=1, sort_keys=True)
print(json.dumps({k: v["in_units_of_C"] for k, v in out["kernels"].items()}, indent=1))
