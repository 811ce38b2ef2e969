#!/bin/bash
# bench.py over the variant libraries of tools/build_variants.sh; prints per-stage ms per step.
# usage: tools/variant_bench.sh out.txt [bench args --] name1 name2 ...   (name "base" = the tree's library)
out=$1; shift
mkdir -p "$(dirname "$out")"
extra=""
while [ "$1" != "--" ] && [ $# -gt 0 ]; do extra="$extra $1"; shift; done
shift
for v in "$@"; do
  if [ "$v" = base ]; then unset PB_LIBPATH; else export PB_LIBPATH=$PWD/build/variants/libpb_$v.so; fi
  line=$(timeout -k 10 120 python bench.py --steps 40 --warmup 5 --regions 1 --no-cpu-baseline $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stage_ms_per_step'], 'alone', d['roofline'].get('alone', {}).get('ms_per_launch'))")
  echo "$v $line" >> $out
done
cat $out
