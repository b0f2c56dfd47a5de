#!/bin/bash
# compile the HIP library with the resource-usage remarks and print VGPR/occupancy per kernel
cd /root/repo/xlab-fftbarotropic_amd || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-value \
  -Rpass-analysis=kernel-resource-usage -o lib/libfftbaro.so csrc/fftbaro.hip csrc/fb_fields.cpp 2> /tmp/build.log
rc=$?
echo rc=$rc
grep -E "error" -A3 /tmp/build.log | head -40
grep -E "Function Name|VGPRs:|Occupancy|ScratchSize" /tmp/build.log | paste - - - - | sed -E 's/remark: [^ ]+ //g; s/\[-Rpass-analysis=kernel-resource-usage\]//g; s/csrc\/[a-z_.]+:[0-9]+:[0-9]+: //g' | awk '{$1=$1};1' | grep -E "${1:-k_}" | cut -c1-170
exit $rc
