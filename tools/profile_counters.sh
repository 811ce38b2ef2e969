#!/bin/bash
# Run ON THE GPU BOX (via gpurun): issue/LDS/instruction-cache counters of the three kernels.
# Usage: tools/profile_counters.sh <tag> [bench args...]  -> gpurun_out/pmc_<tag>/<group>/
# One rocprofv3 run per counter group (they do not fit one pass); kernel trace only, as gpurun requires.
set -e
TAG=$1; shift
export TMPDIR=/tmp
export PB_OVERLAP_DETECT=0     # kernels back to back, so that a counter belongs to one kernel
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
for grp in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU" \
           "SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- python3 bench.py --steps 8 --warmup 4 --regions 1 --no-cpu-baseline --no-residency "$@" > /dev/null 2> $OUT/g$i.err
done
echo counters $TAG
