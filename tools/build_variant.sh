#!/bin/bash
# build a library variant for same-box A/B runs: tools/build_variant.sh <name> [extra hipcc flags...]
# -> xlab-fftbarotropic_amd/lib/alt_<name>.so ; select it with FFTBARO_LIB=<path>
# The exit status is hipcc's; the library appears under its name only after a successful compile (built to a temporary name, renamed),
# so a failed build never leaves -- or leaves in place -- a library that a digest stamp could then declare current.
name=$1; shift
root=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
cd "$root/xlab-fftbarotropic_amd" || exit 1
tmp=lib/alt_$name.so.tmp.$$
log=$(mktemp)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=on -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-value -w "$@" \
  -o "$tmp" csrc/fftbaro.hip csrc/fb_fields.cpp csrc/fb_fieldio.cpp csrc/fb_slab_comm.cpp -ldl > "$log" 2>&1
rc=$?
grep -E "error" -A3 "$log" | head -20
rm -f "$log"
if [ $rc -ne 0 ] || [ ! -s "$tmp" ]; then
  rm -f "$tmp"
  echo "build_variant.sh: hipcc failed (rc $rc); lib/alt_$name.so not replaced" >&2
  [ $rc -ne 0 ] && exit $rc
  exit 1
fi
mv -f "$tmp" lib/alt_$name.so
ls -la lib/alt_$name.so | awk '{print $5, $9}'
