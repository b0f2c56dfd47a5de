#!/bin/bash
# rocprofv3 kernel statistics of the C++ drop-in driver ITSELF (host/barotropic_main.out, not the Python binding): BASELINE.md's stated run
# shortened to 300 steps with a record every 100 (4096^2 Kuo2004, dt = 0.75 s).  Run on the GPU box via gpurun; the summary lands in
# gpurun_out/prof_driver_<tag>/ (kernel_stats.csv + the driver's own [timing] lines) and is copied into profiles/ by hand.
# usage: tools/profile_driver.sh <tag> [npts=4096] [steps=300]
tag=${1:-r04_d}; n=${2:-4096}; steps=${3:-300}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/prof_driver_$tag
work=$(mktemp -d /tmp/drv.XXXXXX)
rm -rf $out && mkdir -p $out $work/input $work/output
H=$GRAFT_REPO_ROOT/xlab-fftbarotropic_amd/host
dt=$(python3 -c "print(3.0 if $n <= 1024 else 3.0 * 1024 / $n)")
(cd $work && $H/makefield.out --kind kuo2004 --npts $n 2> /dev/null) || exit 1
cd $work
# the program itself after "--" (never a shell or env in between: the profiler's library has initialised the GPU by then)
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $out/stats -o p --output-format csv -- $H/barotropic_main.out --npts $n --dt $dt --steps $steps --record-step 100 > $out/stdout.log 2> $out/stderr.log || exit 1
# and once without the profiler: the driver's own figures
$H/barotropic_main.out --npts $n --dt $dt --steps 1000 --record-step 100 > /dev/null 2> $out/plain_stderr.log || exit 1
grep "^\[timing\]" $out/plain_stderr.log > $out/timing.txt
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
rm -rf $out/stats $work
cat $out/timing.txt; head -12 $out/kernel_stats.csv
