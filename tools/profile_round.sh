#!/bin/bash
# Profiles of the default bench.py workload for profiles/: kernel-trace statistics, then the HBM traffic counters in
# two passes of their own (FETCH_SIZE and WRITE_SIZE do not fit one pass).  Run on the GPU box from the repo root:
#   bash tools/profile_round.sh r1        -> gpurun_out/prof_r1/...
set -u
TAG=${1:-r1}
R=$(pwd)
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --cpu_n 0 --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/kt.err || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --cpu_n 0 --no-extras --no-timers > $OUT/fetch.log 2>&1 || exit 2
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --cpu_n 0 --no-extras --no-timers > $OUT/write.log 2>&1 || exit 3
python3 $R/tools/profile_summary.py $OUT $TAG
