#!/usr/bin/env python3
"""Timing experiment: per-wave work / barrier-wait cycles of one k_detect2 workgroup (excised stream).
Needs a variant library built with -DD2_STAMP (tools/build_variants.sh) selected by PB_LIBPATH."""
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import synth_second

lp = importlib.import_module("vlite-fast_amd.libpb")
L = lp.load()
dev = torch.device("cuda", 0)
S = 10
h = lp.PbHandle(device=0, nant=1, nbit=8, npol=1, rfi_mode=2, rows_per_seg=1024, max_seg=S, nsets=1)
sec = synth_second(torch, dev, 42, h.seg_samples, S)
torch.cuda.synchronize()
for s in range(S):
    h.submit_planar_dev(0, s, sec[s][0].data_ptr(), sec[s][1].data_ptr(), h.seg_samples)
for _ in range(5):
    h.process(S)
    h.sync()
out = (C.c_ulonglong * 32)()
rc = L.pb_internal_d2_stamps(out)
names = {0: "A", int(os.environ.get("D2_WAVE_L", "5")): "L", 1: "B0", 2: "B1", 3: "B2", 9 - int(os.environ.get("D2_WAVE_L", "5")): "B3"}
nstep = 10 * 1024 // 32 + 2
print("rc", rc, "steps", nstep)
for w in range(6):
    work, wait, extra = out[w * 4], out[w * 4 + 1], out[w * 4 + 2]
    print("wave %d %-3s work %8d (%6.0f/step)  barrier wait %8d (%6.0f/step)  dma wait %8d (%6.0f/step)"
          % (w, names.get(w, "?"), work, work / nstep, wait, wait / nstep, extra, extra / nstep))
h.close()
