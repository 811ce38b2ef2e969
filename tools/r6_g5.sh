set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# issue counters of k_detect3 (variant library) and k_detect2 (shipped), kernels back to back
for v in d3 base; do
  if [ $v = base ]; then unset PB_LIBPATH; else export PB_LIBPATH=$PWD/build/variants/libpb_$v.so; fi
  tools/profile_counters.sh cnt_$v --no-extras --no-power
  python tools/summarise_counters.py cnt_$v 1 > gpurun_out/r6/counters_$v.json
  rm -rf gpurun_out/pmc_cnt_$v
done
python - <<'PY'
import json
for v in ("d3", "base"):
    d = json.load(open("gpurun_out/r6/counters_%s.json" % v))
    for k, c in d["kernels"].items():
        if k.startswith("k_detect"):
            print(v, k, {n: round(x) for n, x in sorted(c.items())})
PY
