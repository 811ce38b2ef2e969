#!/usr/bin/env python3
"""Timing experiment: clock at the phase boundaries of every k_channelize_kur workgroup (one FFT row, both pols).
Needs a variant library built with -DKUR_STAMP (tools/build_variants.sh k_channelize.hip kst="-DKUR_STAMP")
selected by PB_LIBPATH.  Prints the mean phase durations (s_memtime ticks)."""
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
ms = h.timers()["channelize"][0]
NWG = S * 1024
buf = np.zeros((20480, 8), dtype=np.uint64)
L.pb_internal_kur_stamps.argtypes = [C.c_void_p]
rc = L.pb_internal_kur_stamps(buf.ctypes.data)
t = buf[:NWG].astype(np.int64)
print("rc", rc, "k_channelize_kur %.4f ms per launch alone, %d workgroups" % (ms, NWG))
names = ["both rows requested -> staged in LDS (barrier)", "moments of the 50 blocks (barrier)",
         "D'Agostino scores, 50 lanes (barrier)", "flags, ballot, weight, broadcast (2 barriers)",
         "transforms of pol 0 and pol 1"]
for label, sel in (("rows without flags (2 transforms)", t[:, 6] == 0), ("rows with flags (4 transforms)", t[:, 6] != 0)):
    d = np.diff(t[sel, :6], axis=1).astype(np.float64)
    tot = d.sum(axis=1).mean()
    print("%s: %d workgroups, %.0f ticks per workgroup" % (label, sel.sum(), tot))
    for i in range(5):
        print("  %-52s %8.0f  %5.1f %%   (p10 %6.0f  p90 %6.0f)" % (names[i], d[:, i].mean(), 100.0 * d[:, i].mean() / tot,
                                                                 np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
life = (t[:, 5] - t[:, 0]).sum()
span = t[:, 5].max() - t[:, 0].min()
print("sum of workgroup lifetimes / (256 CUs x launch span) = %.2f workgroups resident per CU; span %d ticks = %.1f ticks per us"
      % (life / 256.0 / span, span, span / (ms * 1e3)))
h.close()
