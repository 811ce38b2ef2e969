# taps = 4: the channeliser that flags its own rows (PB_FUSE_KURTOSIS=2) against kurtosis pass + channeliser (=1),
# alternating on one box; then timing experiments of the fused kernel (results invalid with PB_PFB_DBG)
run() { python bench.py --taps 4 --steps 40 --warmup 5 --regions 5 --no-extras --no-cpu-baseline --no-power 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['ms_per_step'], d['value'], d['timed_regions']['ms_per_step_min'], d['timed_regions']['ms_per_step_max'], d['stage_ms_per_step'])"; }
for v in 2 1; do echo "== PB_FUSE_KURTOSIS=$v"; PB_FUSE_KURTOSIS=$v run; done
echo "== unfused, kurtosis launch left out once the flags exist (PB_SKIP=4)"; PB_FUSE_KURTOSIS=1 PB_SKIP=4 run
export PB_FUSE_KURTOSIS=2
echo "== fused, own flags from the previous launch's bytes: no staging of the own row, no moments (PB_PFB_DBG=1)"; PB_PFB_DBG=1 run
echo "== fused, no look-back wait: predecessors' words by ordinary loads (PB_PFB_DBG=2)"; PB_PFB_DBG=2 run
echo "== fused, both (PB_PFB_DBG=3)"; PB_PFB_DBG=3 run
