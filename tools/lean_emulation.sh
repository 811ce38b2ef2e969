# Occupancy emulation (results INVALID: LDS accesses beyond the allocation are dropped): the 128-register channeliser
# (-DFFT_LEAN, two-kernel path) launched with less LDS than its exchange needs, so that the dispatcher packs it the
# way a split-exchange kernel of that footprint would be packed -- what would 3 channeliser workgroups + 1 detect
# workgroup per CU be worth to the pipelined step?   needs: tools/build_variants.sh k_channelize.hip lean="-DFFT_LEAN"
# DEPENDS ON gfx950's LDS bounds behaviour (out-of-range LDS accesses of a wave are dropped / return 0, no fault) and must
# run ONLY with the FFT_LEAN variant build (the one that reads PB_LEAN_LDS): never point it at another library.
run() { python bench.py --steps 40 --warmup 5 --regions 3 --no-extras --no-cpu-baseline --no-power 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['ms_per_step'], d['timed_regions']['ms_per_step_min'], d['timed_regions']['ms_per_step_max'], d['stage_ms_per_step'], 'alone', d['roofline'].get('alone', {}).get('ms_per_launch'))"; }
export PB_FUSE_KURTOSIS=0
for i in 1 2; do
echo "== shipped library, two kernels"; (unset PB_LIBPATH; run)
export PB_LIBPATH=$PWD/build/variants/libpb_lean.so
for lds in 50000 40000 30000 25000; do echo "== lean (128 VGPRs), $lds B of LDS"; PB_LEAN_LDS=$lds run; done
done
