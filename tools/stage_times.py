#!/usr/bin/env python3
"""Kernels of one second of data back to back on ONE stream (nsets = 1): hipEvent time of each stage alone."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import synth_second

lp = importlib.import_module("vlite-fast_amd.libpb")
dev = torch.device("cuda", 0)
S = 10
taps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
h = lp.PbHandle(device=0, nant=1, nbit=8, npol=1, rfi_mode=2, rows_per_seg=1024, max_seg=S, nsets=1, taps=taps)
sec = synth_second(torch, dev, 42, h.seg_samples, S)
torch.cuda.synchronize()
for s in range(S):
    h.submit_planar_dev(0, s, sec[s][0].data_ptr(), sec[s][1].data_ptr(), h.seg_samples)
h.profile(True)
for _ in range(5):
    h.process(S)
    h.sync()
h.timers(reset=True)
N = 20
for _ in range(N):
    h.process(S)
    h.sync()
print("taps %d, ms per launch alone:" % taps, {k: round(v[0] / v[1], 4) for k, v in h.timers().items() if v[1]})
h.close()
