#!/bin/bash
# SQ counter passes over a short bench run (4096^2 unless a grid is given); summary -> gpurun_out/pmc_sq.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
G=${1:-4096}
rm -rf gpurun_out/pmc_sq1 gpurun_out/pmc_sq2
timeout -k 10 250 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace -d gpurun_out/pmc_sq1 -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-steps 0 --driver-steps 0 --grid $G > gpurun_out/pmc1.log 2>&1 &&
timeout -k 10 250 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU --kernel-trace -d gpurun_out/pmc_sq2 -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-steps 0 --driver-steps 0 --grid $G > gpurun_out/pmc2.log 2>&1 &&
python3 tools/pmc_summary.py gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 > gpurun_out/pmc_sq.json
