#!/usr/bin/env python3
"""Timing experiment: which workgroups of k_channelize_kur and k_detect2 sit on which CU while the pipeline runs as
benchmarked (three buffer sets).  Needs a library built with -DKUR_STAMP (k_channelize.hip) AND -DD2_STAMP
(k_detect2.hip), selected by PB_LIBPATH.  Prints, at ten instants of one mid-pipeline channeliser launch, how many CUs
hold (c channeliser, d detect) workgroups."""
import collections
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
S, NSETS, NL = 10, 3, 24
h = lp.PbHandle(device=0, nant=1, nbit=8, npol=1, rfi_mode=2, rows_per_seg=1024, max_seg=S, nsets=NSETS)
sec = synth_second(torch, dev, 42, h.seg_samples, S)
torch.cuda.synchronize()
for st in range(NSETS):
    h.select_set(st)
    for s in range(S):
        h.submit_planar_dev(0, s, sec[s][0].data_ptr(), sec[s][1].data_ptr(), h.seg_samples)
h.sync()
for k in range(NL):
    h.select_set(k % NSETS)
    h.process(S)
    if k >= 2:
        h.select_set((k - 2) % NSETS)
        h.fetch_view(0, 1, S)
h.sync()
kb = np.zeros((4, 10240, 10), dtype=np.uint64)
L.pb_internal_kur_stamps.argtypes = [C.c_void_p]
assert L.pb_internal_kur_stamps(kb.ctypes.data) == 0
db = np.zeros((64, 256, 4), dtype=np.uint64)
L.pb_internal_d2_wg.argtypes = [C.c_void_p]
assert L.pb_internal_d2_wg(db.ctypes.data) == 0
ch = kb[(NL - 3) & 3].astype(np.int64)            # a mid-pipeline channeliser launch
c0, c1 = ch[:, 8], ch[:, 9]


def cu_key(v):
    xcc, hw = v & 0xff, v >> 8
    return (xcc & 0xf) * 1024 + ((hw >> 13) & 7) * 128 + ((hw >> 12) & 1) * 64 + ((hw >> 8) & 0xf)


ckey = cu_key(ch[:, 7])
d = db.astype(np.int64).reshape(-1, 4)
d = d[d[:, 1] > 0]
dkey = cu_key(d[:, 2])
print("channeliser launch: %.1f us, %d distinct CUs; detect records: %d workgroups on %d distinct CUs"
      % ((c1.max() - c0.min()) / 100.0, len(set(ckey.tolist())), len(d), len(set(dkey.tolist()))))
cus = sorted(set(ckey.tolist()))
for frac in (0.05, 0.15, 0.25, 0.35, 0.45, 0.55, 0.65, 0.75, 0.85):
    t = c0.min() + frac * (c1.max() - c0.min())
    cc = collections.Counter(ckey[(c0 <= t) & (c1 > t)].tolist())
    dd = collections.Counter(dkey[(d[:, 0] <= t) & (d[:, 1] > t)].tolist())
    hist = collections.Counter((cc.get(k, 0), dd.get(k, 0)) for k in cus)
    print("%3.0f %% of the launch: %4d channeliser + %3d detect workgroups resident; CUs by (channeliser, detect): %s"
          % (100 * frac, sum(cc.values()), sum(dd.values()), ", ".join("%s x%d" % (k, v) for k, v in sorted(hist.items()))))
h.close()
