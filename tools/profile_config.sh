#!/bin/bash
# rocprofv3 summaries of one tools/phase_times.py workload for profiles/: kernel-trace statistics, and (with "pmc" as
# second argument) the HBM traffic counters in two passes of their own.  Run on the GPU box from the repo root:
#   bash tools/profile_config.sh config5 pmc --kind random --sigma 255 --n 1000000000 --leaves 2
set -u
TAG=$1; PMC=$2; shift 2
R=$(pwd)
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python3 $R/tools/phase_times.py "$@" > $OUT/phase_times.txt 2> $OUT/kt.err || exit 1
cp $OUT/kt/kt_kernel_stats.csv $R/gpurun_out/r3_${TAG}_kernel_stats.csv
cp $OUT/phase_times.txt $R/gpurun_out/r3_${TAG}_phase_times.txt
if [ "$PMC" = "pmc" ]; then
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- python3 $R/tools/phase_times.py "$@" > $OUT/fetch.log 2>&1 || exit 2
  timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- python3 $R/tools/phase_times.py "$@" > $OUT/write.log 2>&1 || exit 3
  { echo "rocprofv3 --pmc FETCH_SIZE, then --pmc WRITE_SIZE (separate passes) of: tools/phase_times.py $*"; echo "per dispatch averages as the counters report them (in KB; a 128-byte read request is tallied as 64 bytes on gfx950):"; python3 $R/tools/pmc_summary.py $OUT/fetch/f_counter_collection.csv; python3 $R/tools/pmc_summary.py $OUT/write/w_counter_collection.csv; } > $R/gpurun_out/r3_${TAG}_pmc_traffic.txt
fi
