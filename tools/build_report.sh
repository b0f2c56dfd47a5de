#!/bin/bash
# compile the HIP library with the resource-usage remarks and print VGPR/occupancy per kernel.
# Builds into a scratch .so (never the live lib/libfftbaro.so); optional arg: kernel-name regex.
root=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
cd "$root/xlab-fftbarotropic_amd" || exit 1
log=${TMPDIR:-/tmp}/fb_build_report.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=on -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-value \
  -Rpass-analysis=kernel-resource-usage -o "${TMPDIR:-/tmp}/fb_build_report.so" csrc/fftbaro.hip csrc/fb_fields.cpp csrc/fb_fieldio.cpp csrc/fb_slab_comm.cpp -ldl $FB_EXTRA_FLAGS 2> "$log"
rc=$?
echo rc=$rc
grep -E "error" -A3 "$log" | head -40
grep -E "Function Name|VGPRs:|Occupancy|ScratchSize|LDS Size" "$log" | paste - - - - - | sed -E 's/remark: [^ ]+ //g; s/\[-Rpass-analysis=kernel-resource-usage\]//g; s/csrc\/[a-z_0-9.]+:[0-9]+:[0-9]+: //g' | awk '{$1=$1};1' | grep -E "${1:-k_}" | cut -c1-200
exit $rc
