#!/bin/bash
# Same-box A/B of the drop-in driver's record path: one set of pinned record buffers (round 3: the step loop waits for the previous record's
# files at every record step) against two (round 4), interleaved.  BASELINE.md's stated run: 4096^2 Kuo2004, 1000 steps, a record every 100.
# usage (on the GPU box): tools/driver_record_ab.sh [pairs=3]   -> gpurun_out/driver_record_ab.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
H=$GRAFT_REPO_ROOT/xlab-fftbarotropic_amd/host
work=$(mktemp -d /tmp/drvab.XXXXXX); mkdir -p $work/input $work/output
(cd $work && $H/makefield.out --kind kuo2004 --npts 4096 2> /dev/null) || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/driver_record_ab.txt; : > $out
cd $work
for i in $(seq 1 ${1:-3}); do
  for b in 1 2; do
    $H/barotropic_main.out --npts 4096 --dt 0.75 --steps 1000 --record-step 100 --record-buffers $b > /dev/null 2> err.log || exit 1
    python3 - "$b" >> $out <<'PY'
import re, sys
t = open("err.log").read()
c = re.search(r"without the record steps ([0-9.]+) steps/s", t).group(1)
w = re.search(r"\): ([0-9.]+) steps/s over", t).group(1)
r = re.search(r"([0-9.]+) s with a record step holding", t).group(1)
g = re.search(r"= ([0-9.]+) GB/s to", t).group(1)
s = re.search(r"slowest record ([0-9.]+) s to write, a stretch between records is ([0-9.]+) s", t).groups()
tail, un = re.search(r"([0-9.]+) s between the last step's end and the last file \(writer tail\), (-?[0-9.]+) s unaccounted", t).groups()
hw = re.search(r"branches: ([0-9.]+) s waiting for a free set", t).group(1)
print("record-buffers %s: compute-only %s steps/s, with records %s steps/s (%.1f %% less); record steps held the compute stream %s s, writer tail %s s, unaccounted %s s; "
      "writer %s GB/s, slowest record %s s (stretch %s s); host waited %s s for a buffer set"
      % (sys.argv[1], c, w, 100 * (1 - float(w) / float(c)), r, tail, un, g, s[0], s[1], hw))
PY
  done
done
rm -rf $work
cat $out
