#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel-trace statistics and HBM traffic counters for bench.py.
# Usage: tools/profile_gpu.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/{stats,fetch,write}
# Counters are collected in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -e
TAG=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 40 --warmup 10 --regions 1 --no-cpu-baseline --no-residency "$@" > $OUT/bench_stats.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 8 --warmup 4 --regions 1 --no-cpu-baseline --no-residency "$@" > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 8 --warmup 4 --regions 1 --no-cpu-baseline --no-residency "$@" > /dev/null 2> $OUT/write.err
echo profiled $TAG
