set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r6/t_all1.log 2>&1 || { tail -40 gpurun_out/r6/t_all1.log; exit 1; }
tail -3 gpurun_out/r6/t_all1.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r6/bench_driver_cmd_a.json 2> gpurun_out/r6/bench_driver_cmd_a.err
python -c "
import json;d=json.load(open('gpurun_out/r6/bench_driver_cmd_a.json'));print(d['ms_per_step'],d['ms_per_step_cold'],d['taps4']['ms_per_step'],d['taps4']['ms_per_step_cold'],d['roofline']['power'].get('gfx_mhz'), d['roofline']['residency'], d['roofline']['survey_model'])"
