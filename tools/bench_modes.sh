#!/bin/bash
# Run ON THE GPU BOX: one bench line per secondary mode -> gpurun_out/r3/modes.jsonl
out=${1:-gpurun_out/r3/modes.jsonl}; mkdir -p $(dirname $out); rm -f $out
run() { echo "== $*" >&2; timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 5 --regions 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); d['args']='$*'; print(json.dumps(d))" >> $out; }
run
run --taps 4
run --nbit 2
run --rfi-mode 1
run --rfi-mode 0
run --rfi-frac 0.01
run --rfi-frac 0.3
run --ant-per-gpu 8 --steps 12 --warmup 3
run --ant-per-gpu 2
run --backend hipfft --steps 10 --warmup 2
python - "$out" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print("%-44s %8.4f ms/step  %8.1fx per antenna  %9.1f Msamp/s aggregate  frac %.3f" % (d['args'] or '(default)', d['ms_per_step'], d['x_realtime_per_antenna'], d['value'], d['roofline']['frac']))
PY
