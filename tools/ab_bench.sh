#!/bin/bash
# same-box A/B: tools/ab_bench.sh <grid> <variant[:ENV=VAL[,ENV=VAL]]...>   ("base" = the in-tree library)
n=$1; shift
for spec in "$@"; do
  v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=$(echo "${spec#*:}" | tr ',' ' ')
  if [ "$v" = base ]; then lib=""; else lib="FFTBARO_LIB=$GRAFT_REPO_ROOT/xlab-fftbarotropic_amd/lib/alt_$v.so"; fi
  env $lib $envs timeout -k 10 200 python bench.py --steps 10 --warmup 3 --cpu-steps 0 --driver-steps 0 --grid $n 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$spec', d['config']['grid'][0], round(d['value'],2), {k: round(v,4) for k,v in d['kernels_ms_per_launch'].items()})"
done
