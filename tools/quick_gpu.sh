#!/bin/bash
# quick GPU check: model parity tests (unless SKIP_TESTS=1) + bench lines at the grids given (default 4096 8192 2048)
if [ -z "$SKIP_TESTS" ]; then python -m pytest tests/test_gpu_parity.py -x -q -k "model or step or nonsquare" 2>&1 | tail -2; fi
for n in ${@:-4096 8192 2048}; do timeout -k 10 200 python bench.py --steps 10 --warmup 3 --cpu-steps 0 --grid $n 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['grid'], round(d['value'],2), {k: round(v,4) for k,v in d['kernels_ms_per_launch'].items()})"; done
