# two antennas per GPU (configs[3]'s per-GPU load): one detect launch for both antennas against one launch per antenna
# back to back (PB_DETECT_SERIAL=1), detect's ring two / three chunks deep; alternating, same box
run() { python bench.py --steps 40 --warmup 5 --regions 3 --no-extras --no-cpu-baseline --no-power --ant-per-gpu ${A:-2} 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['ms_per_step'], d['x_realtime_per_antenna'], d['timed_regions']['ms_per_step_min'], d['timed_regions']['ms_per_step_max'], d['stage_ms_per_step'])"; }
for i in 1 2; do
echo "== one launch (grid z = A)"; PB_DETECT_SERIAL=0 run
echo "== per antenna, depth 2"; PB_DETECT_SERIAL=1 PB_DETECT_DEPTH=2 run
echo "== per antenna, depth 3"; PB_DETECT_SERIAL=1 PB_DETECT_DEPTH=3 run
done
