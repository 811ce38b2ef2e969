#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the whole profile of one configuration -- kernel-trace statistics, HBM traffic
# (FETCH_SIZE / WRITE_SIZE passes) and the issue counters -- summarised on the box into gpurun_out/profiles_<tag>/
# (the raw rocprofv3 output is too large to travel back: deleted here).  Copy the summaries into profiles/ afterwards.
# Usage: tools/profile_round.sh <tag> <round> [bench args...]      e.g. tools/profile_round.sh r04a r04
#        taps = 4: tools/profile_round.sh pfb r04 --taps 4
set -e
TAG=$1; RND=$2; shift; shift
TAPS=1
for a in "$@"; do if [ "$prev" = "--taps" ]; then TAPS=$a; fi; prev=$a; done
cd $GRAFT_REPO_ROOT
tools/profile_gpu.sh $TAG --no-extras --no-power "$@"
python tools/summarise_profile.py $TAG $RND
rm -rf gpurun_out/prof_$TAG/stats gpurun_out/prof_$TAG/fetch gpurun_out/prof_$TAG/write
tools/profile_counters.sh $TAG --no-extras --no-power "$@"
python tools/summarise_counters.py $TAG $TAPS > profiles/${RND}_${TAG}_issue_counters.json
rm -rf gpurun_out/pmc_$TAG
mkdir -p gpurun_out/profiles_$TAG
cp profiles/${RND}_${TAG}_* gpurun_out/profiles_$TAG/
ls -la gpurun_out/profiles_$TAG
