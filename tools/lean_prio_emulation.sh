# Occupancy emulation, continued (results INVALID, see tools/lean_emulation.sh: run ONLY with the FFT_LEAN variant builds):
# the 128-register channeliser launched with 25 / 30 KB of LDS packs three or four to a CU beside a detect workgroup;
# round 5 found detect then takes as much longer as the channeliser gains.  Round 6: (a) all of detect's waves at wave
# priority 1 / 3 (leanp1 / leanp3), (b) the channeliser started 10 / 20 us late so that detect's workgroups land first
# (leand10 / leand20).   needs: tools/build_variant_multi.sh lean "-DFFT_LEAN" k_channelize.hip
#   tools/build_variant_multi.sh leanp3 "-DFFT_LEAN -DD2_PRIO_B=3" k_channelize.hip k_detect2.hip
#   tools/build_variant_multi.sh leand10 "-DFFT_LEAN -DPB_CHAN_DELAY_US=10" k_channelize.hip pb_api.hip
run() { python bench.py --steps 40 --warmup 5 --regions 3 --no-extras --no-cpu-baseline --no-power --no-residency 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['ms_per_step'], d['timed_regions']['ms_per_step_min'], d['timed_regions']['ms_per_step_max'], d['stage_ms_per_step'], 'alone', d['roofline'].get('alone', {}).get('ms_per_launch'))"; }
export PB_FUSE_KURTOSIS=0
for i in 1 2; do
echo "== shipped library, two kernels"; (unset PB_LIBPATH; run)
for v in ${VARIANTS:-lean leand10 leand20}; do
  export PB_LIBPATH=$PWD/build/variants/libpb_$v.so
  for lds in 50000 30000 25000; do echo "== $v, $lds B of LDS"; PB_LEAN_LDS=$lds run; done
done
done
