# Run ON THE GPU BOX: the round's records with the final kernels.  usage: bash tools/r6_final.sh [part ...]
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6f
mkdir -p $O
parts=${@:-"tests bench rehearse probes profile"}
for part in $parts; do case $part in
tests)
  timeout -k 10 900 python -m pytest tests -m gpu -q > $O/t_all.log 2>&1 || { tail -40 $O/t_all.log; exit 1; }
  tail -2 $O/t_all.log ;;
bench)
  timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
  timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --ant-per-gpu 2 --no-cpu-baseline --no-extras > $O/bench_two_antennas.json 2> $O/bench_two_antennas.err
  python - <<'PY'
import json
for f in ("bench_driver_cmd", "bench", "bench_two_antennas"):
    d = json.load(open("gpurun_out/r6f/%s.json" % f))
    print(f, d["ms_per_step"], d["ms_per_step_cold"], d.get("taps4", {}).get("ms_per_step"), d.get("taps4", {}).get("ms_per_step_cold"),
          d["roofline"].get("power", {}).get("gfx_mhz"), (d["roofline"].get("residency") or {}).get("measured_in_run"))
PY
  ;;
rehearse)
  # configs[3]'s real shape on the one card: eight ranks as threads of one process (the pool allows six PROCESSES on a card)
  timeout -k 10 600 python bench.py --gpus 8 --share-gpus --dist-backend threads --steps 5 --warmup 2 > $O/threads8_configs3_rehearsal.json 2> $O/threads8.err || { tail -20 $O/threads8.err; exit 1; }
  timeout -k 10 600 python bench.py --gpus 8 --share-gpus --dist-backend threads --coadd-layout root --steps 5 --warmup 2 --no-extras > $O/threads8_root_layout.json 2> $O/threads8_root.err
  PB_BENCH_CONFIGS3_AT=4 timeout -k 10 600 python bench.py --gpus 4 --share-gpus --dist-backend gloo --steps 5 --warmup 2 > $O/gloo4_configs3_rehearsal.json 2> $O/gloo4.err
  timeout -k 10 600 python bench.py --gpus 2 --share-gpus --dist-backend gloo --steps 5 --warmup 2 > $O/gloo2_launcher.json 2> $O/gloo2.err
  python - <<'PY'
import json
for f in ("threads8_configs3_rehearsal", "threads8_root_layout", "gloo4_configs3_rehearsal", "gloo2_launcher"):
    d = json.load(open("gpurun_out/r6f/%s.json" % f))
    print(f, d["n_gpus"], d.get("rccl_ranks"), d.get("dist_backend"), d["coadd_order"]["layout"], d["config"]["antennas"], "configs3" in d and d["configs3"]["antennas"])
PY
  ;;
probes)
  export PB_LIBPATH=$PWD/vlite-fast_amd/csrc/libpb_hip_exp.so
  (PB_SKIP=2 python tools/corun_probe.py; PB_SKIP=0 python tools/corun_probe.py none) > $O/corun_probe.txt 2>&1
  for skip in 0 2 1 0; do PB_SKIP=$skip python tools/energy_probe.py 2.5 >> $O/energy.txt 2>&1; done
  PB_SKIP=0 python tools/energy_probe.py 2.5 4 >> $O/energy.txt 2>&1
  unset PB_LIBPATH
  tail -6 $O/energy.txt ;;
profile)
  tools/profile_round.sh r06a r06
  tools/profile_round.sh pfb r06 --taps 4 ;;
esac; done
