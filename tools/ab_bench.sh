#!/bin/bash
# Same-box A/B of library variants: bench.py (short regions, no extras) over build/variants/libpb_<name>.so and the
# tree's library ("base"), ALTERNATING over `reps` rounds so that clock / box drift hits every variant alike.
# usage: tools/ab_bench.sh out.txt reps "bench args" name1 name2 ...       (bench args e.g. "--taps 4" or "--ant-per-gpu 2")
out=$1; reps=$2; extra=$3; shift 3
mkdir -p "$(dirname "$out")"
echo "# bench.py --steps 40 --warmup 5 --regions 3 --no-extras --no-cpu-baseline --no-power $extra ; variants: $*" >> $out
for rep in $(seq 1 $reps); do
  for v in "$@"; do
    if [ "$v" = base ]; then unset PB_LIBPATH; else export PB_LIBPATH=$PWD/build/variants/libpb_$v.so; fi
    line=$(timeout -k 10 180 python bench.py --steps 40 --warmup 5 --regions 3 --no-extras --no-cpu-baseline --no-power $extra 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r=d['timed_regions']
print('step %.4f (%.4f - %.4f) cold %.4f  in-pipeline %s  alone %s' % (d['ms_per_step'], r['ms_per_step_min'], r['ms_per_step_max'], d['ms_per_step_cold'], d['stage_ms_per_step'], d['roofline'].get('alone', {}).get('ms_per_launch')))")
    echo "rep $rep  $v  $line" | tee -a $out
  done
done
