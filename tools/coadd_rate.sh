#!/bin/bash
# Run ON THE GPU BOX: end-to-end rate of the coadder host (BASELINE configs[3]'s product) on full-size genbase dumps:
# A antenna dumps -> coadd_host.py (one rank, A antennas batched; or --ranks 2 with gloo on the one GPU) -> every
# antenna's .fil / _kur.fil + ONE coadded _ea99_kur.fil.   Usage: tools/coadd_rate.sh <antennas> <seconds> [extra args]
set -e
A=${1:-2}; T=${2:-6}; shift; shift || true
D=/tmp/coadd_rate; rm -rf $D; mkdir -p $D
files=""
for a in $(seq 0 $((A-1))); do
  python -m vlite-fast_amd.genbase -t $T -r $((42+a)) -d 2 -p 0.05 --unix-time 1467334800 --out $D/ant$a.vdif > $D/genbase$a.log 2>&1
  files="$files $D/ant$a.vdif"
done
ls -la $D/*.vdif
S=$(date +%s.%N); python vlite-fast_amd/coadd_host.py --replay $files -w 2 -b 8 -r 2 -o --datadir $D --logdir $D "$@" > $D/co.log 2> $D/co.err || { tail -5 $D/co.log; tail -20 $D/co.err; exit 1; }
E=$(date +%s.%N); python3 -c "print('coadd_host wall %.2f s for %s antennas x %s s of data in (incl. python + torch start-up)' % ($E - $S, '$A', '$T'))"
grep -h "Proc Time\|Wrote" $D/co.log | tail -4
ls -la $D/*.fil
