#!/bin/bash
# build a library variant for same-box A/B runs: tools/build_variant.sh <name> [extra hipcc flags...]
# -> xlab-fftbarotropic_amd/lib/alt_<name>.so ; select it with FFTBARO_LIB=<path>
name=$1; shift
root=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
cd "$root/xlab-fftbarotropic_amd" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=on -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-value -w "$@" \
  -o lib/alt_$name.so csrc/fftbaro.hip csrc/fb_fields.cpp csrc/fb_fieldio.cpp csrc/fb_slab_comm.cpp -ldl 2>&1 | grep -E "error" -A3 | head -20
ls -la lib/alt_$name.so | awk '{print $5, $9}'
