#!/bin/bash
# Round-4 experiment: k_rowh2 as two 512-thread workgroups per x2 (fb_rowh.h, k_rowh2s; -DRH2_SPLIT_EXPERIMENT builds
# from tools/build_variant.sh rh2s / rh2s0).  Same-box A/B of the 8192^2 row pass + the fetch counters of both.  Run on the GPU box.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-rh2s}; mkdir -p $out
{
tools/ab_bench.sh 8192 base rh2s:FB_ROWH2_SPLIT=1 rh2s:FB_ROWH2_SPLIT=1,FB_ROW_GRID=16384 rh2s0:FB_ROWH2_SPLIT=1 rh2s0:FB_ROWH2_SPLIT=1,FB_ROW_GRID=16384 base rh2s:FB_ROWH2_SPLIT=1
} > $out/ab.txt 2>&1
cat $out/ab.txt
for v in base rh2s; do
  if [ $v = base ]; then E=""; else export FFTBARO_LIB=$GRAFT_REPO_ROOT/xlab-fftbarotropic_amd/lib/alt_rh2s.so FB_ROWH2_SPLIT=1; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace -d $out/pmc_${v}_$c -o p --output-format csv -- python3 bench.py --grid 8192 --steps 3 --warmup 1 --cpu-steps 0 --driver-steps 0 > $out/pmc_${v}_$c.log 2>&1 || exit 1
  done
  python3 tools/pmc_summary.py $out/pmc_${v}_FETCH_SIZE $out/pmc_${v}_WRITE_SIZE > $out/pmc_$v.json
  rm -rf $out/pmc_${v}_FETCH_SIZE $out/pmc_${v}_WRITE_SIZE
done
python3 - <<PY
import json
for v in ("base", "rh2s"):
    d = json.load(open("$out/pmc_%s.json" % v))
    for k, c in d.items():
        if k.startswith("k_rowh2"):
            print(v, k, "launches", c["launches"], "fetch KB", c.get("FETCH_SIZE"), "write KB", c.get("WRITE_SIZE"), "hbm bytes (2*F+W)*1024 = %.1f MB" % ((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / 1e6))
PY
