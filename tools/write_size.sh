#!/bin/bash
# Run ON THE GPU BOX: WRITE_SIZE / FETCH_SIZE of the step's kernels only (a quick check of a build's HBM traffic)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/ws_$1; shift
rm -rf $OUT; mkdir -p $OUT
for c in WRITE_SIZE FETCH_SIZE; do
rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 bench.py --steps 8 --warmup 4 --regions 1 --no-cpu-baseline --no-extras --no-power "$@" > /dev/null 2> $OUT/$c.err
python3 - $OUT/$c $c <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == sys.argv[2] and "k_" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][:40]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    v = sorted(v)[len(v) // 4:]
    print(sys.argv[2], k, "%.1f MB per launch (x2 for FETCH_SIZE)" % (sum(v) / len(v) * 1024 / 1e6))
PY
rm -rf $OUT/$c
done
