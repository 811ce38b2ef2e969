#!/usr/bin/env python3
"""Timing experiment: amdsmi's view of the headline pipeline -- socket power against the cap, per-XCD gfx clocks and the
accumulated throttle-residency counters of the GPU metrics table before and after a few seconds of steps."""
import importlib, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import amdsmi
import torch
from bench import synth_second
lp = importlib.import_module("vlite-fast_amd.libpb")
amdsmi.amdsmi_init()
hs = amdsmi.amdsmi_get_processor_handles()
print("handles", len(hs))
g = hs[0]
try:
    print("power cap info", amdsmi.amdsmi_get_power_cap_info(g))
except Exception as e:
    print("cap err", e)
def metrics():
    m = amdsmi.amdsmi_get_gpu_metrics_info(g)
    return m
m0 = metrics()
print({k: v for k, v in m0.items() if isinstance(v, (int, float, str))})
print("list fields:", {k: v for k, v in m0.items() if isinstance(v, (list, tuple)) and len(v) <= 16})
dev = torch.device("cuda", 0)
S, NSETS = 10, 3
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
            p = amdsmi.amdsmi_get_power_info(g)
            c = amdsmi.amdsmi_get_clock_info(g, amdsmi.AmdSmiClkType.GFX)
            samples.append((time.time(), p, c))
        except Exception as e:
            samples.append((time.time(), str(e), None))
        time.sleep(0.05)
th = threading.Thread(target=sampler, daemon=True); th.start()
t0 = time.time(); k = 0
while time.time() - t0 < 4.0:
    h.select_set(k % NSETS); h.process(S)
    if k >= 2:
        h.select_set((k - 2) % NSETS); h.fetch_view(0, 1, S)
    k += 1
h.sync(); t1 = time.time(); stop[0] = True; th.join(timeout=5)
m1 = metrics()
print("%d steps, %.4f ms per step" % (k, (t1 - t0) / k * 1e3))
for ts, p, c in samples[len(samples) // 2:len(samples) // 2 + 3]:
    print("t=%.2f" % (ts - t0), p, c)
print("metrics deltas:", {k2: (m1[k2] - m0[k2]) for k2 in m0 if isinstance(m0[k2], (int, float)) and isinstance(m1.get(k2), (int, float)) and m1[k2] != m0[k2]})
print("after:", {k: v for k, v in m1.items() if isinstance(v, (list, tuple)) and len(v) <= 16})
