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
kern = {k: {c: sum(sorted(v)[len(v) // 4:]) / len(sorted(v)[len(v) // 4:]) for c, v in d.items()} for k, d in acc.items()}
sys.path.insert(0, root)
import bench as benchmod
taps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
# share of packed-f32 instructions among a kernel's vector instructions, from the disassembly of the built library
# (static counts: the transforms' straight-line code dominates both)
packed = {}
try:
    import re, subprocess
    csrc = os.path.join(root, "vlite-fast_amd", "csrc")
    flags = ("-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt "
             "-fno-gpu-flush-denormals-to-zero -I../../include -I. -w --cuda-device-only -S").split()
    dis = ""
    for src in ("k_channelize.hip", "k_detect2.hip", "k_channelize_pfb.hip", "k_kurtosis.hip"):
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-o", "-", src], cwd=csrc, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        dis += r.stdout.decode()
    cur = None
    cnt = collections.defaultdict(lambda: [0, 0])
    for line in dis.splitlines():
        m = re.match(r"^(_Z\S+):", line)
        if m:
            cur = m.group(1)
            continue
        t = line.split()
        if cur and len(t) > 0:
            op = t[0]
            if op.startswith("v_") and not op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
                cnt[cur][0] += 1
                if op.startswith("v_pk_"):
                    cnt[cur][1] += 1
    for k in kern:
        for sym, (n, pk) in cnt.items():
            if k in sym and n:
                packed[k] = round(max(packed.get(k, 0.0), pk / n), 3)
except Exception as e:
    packed = {"error": str(e)}
print(json.dumps({"tag": tag, "taps": taps, "kernel_source_sha16": benchmod.kernel_source_hash(), "packed_share": packed,
                  "kernels": kern}, indent=1))
