# needs the experiments build: make -C vlite-fast_amd/csrc exp   (PB_SKIP is not compiled into the shipped library)
run() { python bench.py --steps 40 --warmup 5 --regions 5 --no-extras --no-cpu-baseline --no-power 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['ms_per_step'], d['value'], d['timed_regions']['ms_per_step_min'], d['timed_regions']['ms_per_step_max'], d['stage_ms_per_step'])"; }
for i in 1 2; do
echo "== taps 1 fused (default)"; run
echo "== taps 1 two kernels"; PB_FUSE_KURTOSIS=0 run
echo "== taps 1 two kernels, kurtosis launch left out once flags exist (experiments build: results invalid)"; PB_LIBPATH=$PWD/vlite-fast_amd/csrc/libpb_hip_exp.so PB_FUSE_KURTOSIS=0 PB_SKIP=4 run
done
