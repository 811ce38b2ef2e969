#!/bin/bash
# Run ON THE GPU BOX: end-to-end rate of the reference-shaped executable on a genbase dump
# (VDIF file -> host reader -> H2D -> GPU deframe + kernels -> D2H -> two .fil files).
# Usage: [PB_HOST=<executable>] [PB_EXTRA=-t] tools/replay_rate.sh <seconds of data>   (257.6 MB of VDIF per second; N s in -> N-1 s out)
set -e
T=${1:-4}
D=/tmp/replay_rate; rm -rf $D; mkdir -p $D
python -m vlite-fast_amd.genbase -t $T -r 42 -d 2 -p 0.05 --out $D/dump.vdif > $D/genbase.log 2>&1
ls -la $D/dump.vdif
PB=${PB_HOST:-python -m vlite-fast_amd.process_baseband}     # PB_HOST=vlite-fast_amd/csrc/process_baseband: the native host
S=$(date +%s.%N); $PB --replay $D/dump.vdif -w 2 -b 8 -P 1 -r 2 -o ${PB_EXTRA:-} --datadir $D --logdir $D --no-control > $D/pb.log 2> $D/pb.err || { tail -20 $D/pb.err; exit 1; }
E=$(date +%s.%N); python3 -c "print('process_baseband wall %.2f s for %s s of data in (incl. python + torch start-up)' % ($E - $S, '$T'))"
tail -5 $D/pb.log
ls -la $D/*.fil
