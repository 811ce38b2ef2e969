#!/usr/bin/env python3
"""Timing experiment: clock at the phase boundaries of every k_kurtosis_row workgroup (kernels back to back on
one stream).  Needs a variant built with -DKU_STAMP_ON (tools/build_variants.sh k_kurtosis.hip kstamp="-DKU_STAMP_ON")."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

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
h.profile(True)
for _ in range(4):
    h.timers(reset=True)
    h.process(S)
    h.sync()
ms = h.timers()["kurtosis"][0]
buf = np.zeros((20480, 6), dtype=np.uint64)
L.pb_internal_ku_stamps.argtypes = [C.c_void_p]
L.pb_internal_ku_stamps(buf.ctypes.data)
t = buf[: S * 1024].astype(np.int64)
d = np.diff(t, axis=1).astype(np.float64)
names = ["rows' bytes: 8 16-byte loads, zero-code patch, 8 LDS stores, barrier", "moments of 50 blocks (12-13 per wave), barrier",
         "D'Agostino statistic of 50 blocks (one lane each, fp64), barrier", "flags, barrier", "row weight, mask (one lane)"]
tot = d.sum(axis=1).mean()
print("kurtosis %.4f ms per launch; %.0f ticks per workgroup; sum of lifetimes / (256 CUs) / launch ticks = %.2f resident"
      % (ms, tot, d.sum() / 256.0 / (ms * 1e3 * 2000)))
for i in range(5):
    print("  %-72s %7.0f  %5.1f %%" % (names[i], d[:, i].mean(), 100 * d[:, i].mean() / tot))
h.close()
