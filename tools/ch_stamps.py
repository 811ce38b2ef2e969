#!/usr/bin/env python3
"""Timing experiment: clock at the phase boundaries of every k_channelize workgroup's first transform.
Needs a variant library built with -DCH_STAMP (tools/build_variants.sh k_channelize.hip stamp="-DCH_STAMP")
selected by PB_LIBPATH.  Prints the mean phase durations of unflagged rows and the tick rate (from the span of
one XCD's stamps against the launch's hipEvent time)."""
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
NWG = 2 * S * 1024
buf = np.zeros((40960, 13), dtype=np.uint64)
L.pb_internal_ch_stamps.argtypes = [C.c_void_p]
rc = L.pb_internal_ch_stamps(buf.ctypes.data)
t = buf[:NWG].astype(np.int64)
print("rc", rc, "channelize %.4f ms per launch" % ms)
hw = t[:, 10]
cu = (t[:, 8] & 15) * 4096 + ((hw >> 13) & 7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 15)
ids = np.unique(cu)
print("distinct (xcc, se, sh, cu):", len(ids), " memrealtime span of the launch: %d ticks" % (t[:, 11].max() - t[:, 11].min()))
rates, occ, gaps = [], [], []
for c in ids[:: max(1, len(ids) // 64)]:
    w = t[cu == c]
    w = w[np.argsort(w[:, 0])]
    span = w[:, 7].max() - w[:, 0].min()
    rates.append(span / (ms * 1e3))
    occ.append((w[:, 7] - w[:, 0]).sum() / span)
print("per CU: first start to last end %.0f ticks per us of the launch (min %.0f max %.0f); sum of first-transform "
      "lifetimes / span = %.2f workgroups resident on average" % (np.mean(rates), np.min(rates), np.max(rates), np.mean(occ)))
c = ids[len(ids) // 2]
w = t[cu == c]
w = w[np.argsort(w[:, 0])]
pre = (t[:, 0] - t[:, 12]).astype(np.float64)
print("kernel entry -> first transform's start (flag bytes, row weight): mean %.0f ticks, p10 %.0f, p90 %.0f"
      % (pre.mean(), np.percentile(pre, 10), np.percentile(pre, 90)))
# slot hand-over: a workgroup can only start once an earlier one of its CU has ended
hand = []
for c in ids[:: max(1, len(ids) // 64)]:
    w = t[cu == c]
    w = w[np.argsort(w[:, 12])]
    ends = np.sort(w[:, 7])   # first-transform ends only: a lower bound of the real ends
    for i in range(3, len(w)):
        hand.append(w[i, 12] - ends[i - 3])
hand = np.array(hand, dtype=np.float64)
print("entry of the i-th workgroup of a CU minus the (i-3)-th first-transform end: mean %.0f, p10 %.0f, p50 %.0f, p90 %.0f ticks"
      % (hand.mean(), np.percentile(hand, 10), np.percentile(hand, 50), np.percentile(hand, 90)))
print("one CU, %d workgroups: start, end (ticks from the CU's first start), flagged" % len(w))
for r in w[:24]:
    print("   entry %8d  start %8d  end %8d %s" % (r[12] - w[0, 12], r[0] - w[0, 12], r[7] - w[0, 12], "F" if r[9] else ""))
names = ["", "stage: 16-B global loads -> LDS, barrier", "unpack: 25 ds_read_u16 + cvt, barrier",
         "pass 1: dft25, 25 ds_write_b64, barrier", "pass 2 loads: 25 ds_read_b64, barrier",
         "pass 2: twiddles, dft25, 25 ds_write_b64, barrier", "pass 3: 30 reads, twiddles, dft10, 30 writes, barrier",
         "spectrum: 32 ds_read_b64, split, |X|^2, 8-16 global stores"]
for label, sel in (("rows without flags", t[:, 9] == 0), ("rows with flags, first transform", t[:, 9] != 0)):
    d = np.diff(t[sel, :8], axis=1).astype(np.float64)
    tot = d.sum(axis=1).mean()
    print("%s: %d workgroups, %.0f ticks per transform" % (label, sel.sum(), tot))
    for i in range(1, 8):
        print("  %-62s %8.0f  %5.1f %%   (p10 %6.0f  p90 %6.0f)" % (names[i], d[:, i - 1].mean(), 100.0 * d[:, i - 1].mean() / tot,
                                                             np.percentile(d[:, i - 1], 10), np.percentile(d[:, i - 1], 90)))
life = (t[:, 7] - t[:, 0]).sum()
print("sum of first-transform lifetimes / (256 CUs x 3) = %.0f ticks" % (life / 768.0))
h.close()
