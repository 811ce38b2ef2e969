#!/bin/bash
# Timing experiment: step time per segment against the batch size (planes per batch against the 256-MB Infinity Cache)
out=$1
for cfg in "10 3" "5 3" "4 3" "3 3" "2 3" "2 2" "1 3" "10 2"; do
  set -- $cfg
  line=$(timeout -k 10 200 python bench.py --steps 60 --warmup 5 --regions 3 --no-cpu-baseline --no-extras --seg-per-step $1 --nsets $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stage_ms_per_step'])")
  echo "S=$1 nsets=$2: $line" >> $out
done
cat $out
