set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
export PB_LIBPATH=$PWD/build/variants/libpb_d3.so
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_pipeline.py tests/test_gpu_schedules.py tests/test_gpu_snr.py tests/test_gpu_pfb.py -m gpu -q -x > gpurun_out/r6/t_d3v3.log 2>&1 || { tail -60 gpurun_out/r6/t_d3v3.log; exit 1; }
tail -2 gpurun_out/r6/t_d3v3.log
(D3=1 PB_LIBPATH=$PWD/build/variants/libpb_d3st.so python tools/d2_stamps.py 2>&1 | grep -v "^launch\|amdgpu.ids"; D2_PIPE=1 D3=1 PB_LIBPATH=$PWD/build/variants/libpb_d3st.so python tools/d2_stamps.py 2>&1 | grep -v "^launch\|amdgpu.ids") > gpurun_out/r6/d3v3_stamps.txt
cat gpurun_out/r6/d3v3_stamps.txt
unset PB_LIBPATH
tools/ab_bench.sh gpurun_out/r6/ab_d3v3_taps1.txt 2 "" base d3 d3b d3d6
tools/ab_bench.sh gpurun_out/r6/ab_d3v3_ant2.txt 1 "--ant-per-gpu 2" base d3
