#!/usr/bin/env python3
"""gpurun_out/pmc_<tag>/g*/  ->  per-kernel mean counter values (JSON on stdout)."""
import collections, csv, glob, json, os, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "gpurun_out", "pmc_" + tag, "g*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("k_"):
            acc[k.split("<")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(sorted(v)[len(v) // 4:]) / len(sorted(v)[len(v) // 4:]) for c, v in d.items()} for k, d in acc.items()}
print(json.dumps(out, indent=1))
