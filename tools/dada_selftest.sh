#!/bin/bash
# Site-side self test of the psrdada legs, for a host that HAS psrdada (the build image has none, so this script
# and vlite-fast_amd/csrc/pb_dada_shim.c have never run against the real library: SURVEY 8f-1).
#
#   1. builds the shim against the site's psrdada             make -C vlite-fast_amd/csrc dada PSRDADA=<prefix>
#   2. creates the rings as the reference's scripts do        dada_db -k 40 -b 257638400 -n 8   (scripts/start_dada, start_writer:12)
#                                                             dada_db -k 42 / 46 -r 2 (scripts/start_dada2:13) for -K / -C
#   3. writes a genbase dump into ring 40 and runs the native host on the ring:  process_baseband -k 40 -K 42 -C 46
#      with readers draining 42 and 46 (dada_dbnull), once with block-level reads (default) and once with the
#      reference's ipcio_read (PB_DADA_THREADS=1)
#   4. runs the same dump through --replay (no psrdada involved) and compares the .fil files byte for byte
#   5. prints the ring-fed rate next to the replay rate
#
# usage: tools/dada_selftest.sh <psrdada prefix> [seconds of data = 6] [gpu id = 0]
set -e
PSRDADA=${1:?usage: tools/dada_selftest.sh <psrdada prefix> [seconds] [gpu]}
T=${2:-6}
GPU=${3:-0}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
export PATH=$PSRDADA/bin:$PATH LD_LIBRARY_PATH=$PSRDADA/lib:${LD_LIBRARY_PATH:-}
command -v dada_db > /dev/null || { echo "dada_db not found under $PSRDADA/bin"; exit 2; }
make -s -C vlite-fast_amd/csrc all dada PSRDADA=$PSRDADA
D=$(mktemp -d /tmp/dada_selftest.XXXXXX)
SEC=257638400                                  # 51 200 frames of 5032 bytes: one second, one ring buffer
python3 -m vlite-fast_amd.genbase -t $T -r 42 -d 30 -p 0.25 -f --out $D/dump.vdif > $D/genbase.log 2>&1
PB=vlite-fast_amd/csrc/process_baseband
COMMON="-w 2 -b 8 -P 1 -r 2 -o -g $GPU --no-control"
mkdir -p $D/replay && $PB --replay $D/dump.vdif $COMMON --datadir $D/replay --logdir $D/replay > $D/replay/log 2>&1
for mode in block ipcio; do
    for k in 40 42 46; do dada_db -k $k -d > /dev/null 2>&1 || true; done
    dada_db -k 40 -l -b $SEC -n 8
    dada_db -k 42 -l -b $((10 * 10 * 524288)) -n 4 -r 1
    dada_db -k 46 -l -b 524288 -n 64 -r 1
    dada_dbnull -k 42 -z > $D/null42.log 2>&1 &
    N42=$!
    dada_dbnull -k 46 -z > $D/null46.log 2>&1 &
    N46=$!
    mkdir -p $D/$mode
    [ $mode = ipcio ] && export PB_DADA_THREADS=1 || unset PB_DADA_THREADS
    S=$(date +%s.%N)
    $PB -k 40 -K 42 -C 46 $COMMON -s --datadir $D/$mode --logdir $D/$mode > $D/$mode/log 2>&1 &
    PBPID=$!
    python3 tools/dump_to_ring.py $D/dump.vdif 40
    wait $PBPID
    E=$(date +%s.%N)
    kill $N42 $N46 2> /dev/null || true
    for k in 40 42 46; do dada_db -k $k -d > /dev/null 2>&1 || true; done
    for f in $D/replay/*.fil; do
        cmp "$f" "$D/$mode/$(basename $f)" && echo "$mode: $(basename $f) identical to --replay"
    done
    python3 -c "print('$mode: ring-fed run %.2f s wall for $T s of data (incl. start-up and the writer)' % ($E - $S))"
done
echo "dada_selftest: ok ($D)"
