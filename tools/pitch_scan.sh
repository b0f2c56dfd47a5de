#!/bin/bash
# tools/pitch_scan.sh <grid>: what the pitch autotuner sees (three runs) and the per-kernel times at every candidate pitch
n=${1:-4096}
for i in 1 2 3; do
  FB_TUNE_VERBOSE=1 timeout -k 10 200 python bench.py --steps 10 --warmup 3 --cpu-steps 0 --grid $n 2> gpurun_out/pitch_scan_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('auto', round(d['value'],2), {k: round(v,4) for k,v in d['kernels_ms_per_launch'].items()})"
  grep "fftbaro: pitch" gpurun_out/pitch_scan_err.txt | tr '\n' ';'; echo
done
for k in 0 1 2 3 4 5 6; do
  FB_PITCH_EXTRA=$k timeout -k 10 200 python bench.py --steps 10 --warmup 3 --cpu-steps 0 --grid $n 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('extra=$k', round(d['value'],2), {k: round(v,4) for k,v in d['kernels_ms_per_launch'].items()})"
done
