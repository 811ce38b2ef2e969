# does it pay to let the previous batch's detect place its workgroups before the next channeliser starts?
run() { python bench.py --steps 40 --warmup 5 --regions 3 --no-extras --no-cpu-baseline --no-power --ant-per-gpu ${A:-1} 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['ms_per_step'], d['timed_regions']['ms_per_step_min'], d['timed_regions']['ms_per_step_max'], d['stage_ms_per_step'])"; }
for i in 1 2; do
for us in 0 5 10 20 40; do echo "== PB_CHAN_DELAY_US=$us depth default"; PB_CHAN_DELAY_US=$us run; done
for us in 10 20; do echo "== PB_CHAN_DELAY_US=$us depth 2"; PB_DETECT_DEPTH=2 PB_CHAN_DELAY_US=$us run; done
echo "== delay 0 depth 2"; PB_DETECT_DEPTH=2 run
done
