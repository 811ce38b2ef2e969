run() { python bench.py --steps 40 --warmup 5 --regions 5 --no-extras --no-cpu-baseline --no-power 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['ms_per_step'], d['value'], d['timed_regions']['ms_per_step_min'], d['timed_regions']['ms_per_step_max'], d['stage_ms_per_step'])"; }
for i in 1 2; do
echo "== taps 1 fused (default)"; run
echo "== taps 1 two kernels"; PB_FUSE_KURTOSIS=0 run
echo "== taps 1 two kernels, kurtosis launch left out once flags exist"; PB_FUSE_KURTOSIS=0 PB_SKIP=4 run
done
