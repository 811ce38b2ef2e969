set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
build/lds_granule > gpurun_out/r6/lds_granule.txt 2>&1 || true
cat gpurun_out/r6/lds_granule.txt
# parity of the real variants first (the t2lds builds must be bit-exact)
for v in; do
  PB_LIBPATH=$PWD/build/variants/libpb_$v.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pfb.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r6/t_$v.log 2>&1 || { tail -30 gpurun_out/r6/t_$v.log; exit 1; }
  tail -1 gpurun_out/r6/t_$v.log
done
tools/ab_bench.sh gpurun_out/r6/ab_t2lds_taps1.txt 3 "" base t2lds t2lds96 zconf
tools/ab_bench.sh gpurun_out/r6/ab_t2lds_taps4.txt 3 "--taps 4" base t2lds t2lds96 zconf
tools/ab_bench.sh gpurun_out/r6/ab_dhalf_taps1.txt 2 "" base dhalf
tools/ab_bench.sh gpurun_out/r6/ab_dhalf_ant2.txt 2 "--ant-per-gpu 2" base dhalf t2lds96
