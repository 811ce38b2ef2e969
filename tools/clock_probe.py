#!/usr/bin/env python3
"""Timing experiment: what the GPU's clocks and power do while the headline pipeline runs (rocm-smi sampled from a
thread while the main thread queues steps).  usage: python tools/clock_probe.py [seconds]"""
import importlib
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import synth_second

lp = importlib.import_module("vlite-fast_amd.libpb")
dev = torch.device("cuda", 0)
S, NSETS = 10, 3
dur = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
h = lp.PbHandle(device=0, nant=1, nbit=8, npol=1, rfi_mode=2, rows_per_seg=1024, max_seg=S, nsets=NSETS)
sec = synth_second(torch, dev, 42, h.seg_samples, S)
torch.cuda.synchronize()
for st in range(NSETS):
    h.select_set(st)
    for s in range(S):
        h.submit_planar_dev(0, s, sec[s][0].data_ptr(), sec[s][1].data_ptr(), h.seg_samples)
h.sync()
samples, stop = [], [False]


def sampler():
    while not stop[0]:
        try:
            r = subprocess.run(["rocm-smi", "-d", "0", "-c", "-P", "-t", "-u"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=10)
            samples.append((time.time(), r.stdout.decode()))
        except Exception as e:
            samples.append((time.time(), "error %s" % e))
        time.sleep(0.2)


print(subprocess.run(["rocm-smi", "-d", "0", "-c", "-P", "-t", "-u"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode())
th = threading.Thread(target=sampler, daemon=True)
th.start()
t0 = time.time()
k = 0
while time.time() - t0 < dur:
    h.select_set(k % NSETS)
    h.process(S)
    if k >= 2:
        h.select_set((k - 2) % NSETS)
        h.fetch_view(0, 1, S)
    k += 1
h.sync()
t1 = time.time()
stop[0] = True
th.join(timeout=15)
print("%d steps in %.2f s = %.4f ms per step" % (k, t1 - t0, (t1 - t0) / k * 1e3))
keep = [s for s in samples if t0 + 1.0 < s[0] < t1]
for ts, txt in keep[:: max(1, len(keep) // 4)]:
    print("---- t = %.1f s" % (ts - t0))
    for line in txt.splitlines():
        if any(w in line for w in ("sclk", "mclk", "fclk", "Power", "Temperature", "GPU use", "socclk")):
            print(line)
