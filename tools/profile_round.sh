#!/bin/bash
# Profiles of a bench workload for profiles/<tag>_*_<grid>: run on the GPU box via gpurun, results land in
# gpurun_out/prof_<tag>_<grid>/ and are copied into profiles/ afterwards (tools/profile_collect.py <tag> <grid>).
#   1. rocprofv3 --kernel-trace --stats          -> kernel_stats.csv
#   2. rocprofv3 --pmc FETCH_SIZE  (own pass)    -> counters
#   3. rocprofv3 --pmc WRITE_SIZE  (own pass)    -> counters
#   4. plain bench.py line                       -> bench.json
# usage: tools/profile_round.sh <tag> [grid=4096] [timed steps=20] [cpu steps of the final bench line=5]
tag=${1:-r02_x}; n=${2:-4096}; k=${3:-20}; cpu=${4:-5}
out=gpurun_out/prof_${tag}_$n
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $out && mkdir -p $out
B="python3 bench.py --grid $n --steps $k --warmup 3 --cpu-steps 0 --driver-steps 0"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/stats -o p --output-format csv -- $B > $out/stats.log 2>&1 &&
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch -o p --output-format csv -- python3 bench.py --grid $n --steps 3 --warmup 1 --cpu-steps 0 --driver-steps 0 > $out/fetch.log 2>&1 &&
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write -o p --output-format csv -- python3 bench.py --grid $n --steps 3 --warmup 1 --cpu-steps 0 --driver-steps 0 > $out/write.log 2>&1 &&
timeout -k 10 900 python3 bench.py --grid $n --cpu-steps $cpu > $out/bench.json 2> $out/bench.err
python3 tools/pmc_summary.py $out/fetch $out/write > $out/pmc.json
find $out/fetch -name "*counter_collection.csv" | head -1 | xargs -I{} head -3 {} > $out/pmc_csv_head.txt
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
# the raw traces are large: keep the summaries only
rm -rf $out/stats $out/fetch $out/write
ls -la $out
