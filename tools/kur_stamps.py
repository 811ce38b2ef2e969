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
PIPE = os.environ.get("KUR_PIPE") == "1"     # as benchmarked: three buffer sets, the previous batch's detect beside it
NSETS = 3 if PIPE else 1
h = lp.PbHandle(device=0, nant=1, nbit=8, npol=1, rfi_mode=2, rows_per_seg=1024, max_seg=S, nsets=NSETS)
sec = synth_second(torch, dev, 42, h.seg_samples, S)
torch.cuda.synchronize()
for st in range(NSETS):
    h.select_set(st)
    for s in range(S):
        h.submit_planar_dev(0, s, sec[s][0].data_ptr(), sec[s][1].data_ptr(), h.seg_samples)
h.sync()
h.profile(True)
for k in range(24 if PIPE else 4):      # (NL below)
    if not PIPE or k == 8:
        h.timers(reset=True)        # (reads the stage timers' events: waits for the device)
    h.select_set(k % NSETS)
    h.process(S)
    if not PIPE:
        h.sync()
    elif k >= 2:
        h.select_set((k - 2) % NSETS)
        h.fetch_view(0, 1, S)
h.sync()
tm = h.timers()["channelize"]
ms = tm[0] / max(1, tm[1])
NWG = S * 1024
buf4 = np.zeros((4, 10240, 10), dtype=np.uint64)
L.pb_internal_kur_stamps.argtypes = [C.c_void_p]
rc = L.pb_internal_kur_stamps(buf4.ctypes.data)
NL = 24 if PIPE else 4
# pipelined: the launch before the last two (its neighbours on both sides exist); alone: the last one
t = buf4[(NL - 3 if PIPE else NL - 1) & 3][:NWG].astype(np.int64)
print("rc", rc, "k_channelize_kur %.4f ms per launch %s, %d workgroups (stamps: the last launch)" % (ms, "in the pipeline" if PIPE else "alone", NWG))
names = ["both rows requested -> staged in LDS (barrier)", "moments of the 50 blocks (barrier)",
         "flags in wave 0 only (waves 1-3 unpack pol 0)", "(mask / weight hand-over: no barrier now)   ",
         "transforms of pol 0 and pol 1"]
for label, sel in (("rows without flags (2 transforms)", t[:, 6] == 0), ("rows with flags (4 transforms)", t[:, 6] != 0)):
    d = np.diff(t[sel, :6], axis=1).astype(np.float64)
    tot = d.sum(axis=1).mean()
    print("%s: %d workgroups, %.0f ticks per workgroup" % (label, sel.sum(), tot))
    for i in range(5):
        print("  %-52s %8.0f  %5.1f %%   (p10 %6.0f  p90 %6.0f)" % (names[i], d[:, i].mean(), 100.0 * d[:, i].mean() / tot,
                                                                 np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
rt0, rt1 = t[:, 8], t[:, 9]                      # 100 MHz, common to the chip
span = (rt1.max() - rt0.min()) / 100.0           # us
life = (rt1 - rt0) / 100.0
print("launch span %.1f us; workgroup life mean %.1f us (p10 %.1f, p50 %.1f, p90 %.1f); sum of lives / (256 CUs x span) = %.2f resident per CU"
      % (span, life.mean(), np.percentile(life, 10), np.percentile(life, 50), np.percentile(life, 90), life.sum() / 256.0 / span))
# the shader clock under this load: s_memtime ticks per microsecond of the 100-MHz clock, per workgroup
tpu = np.diff(t[:, [0, 5]], axis=1)[:, 0].astype(np.float64) / np.maximum(life, 1e-3)
print("s_memtime ticks per us of s_memrealtime: mean %.0f (p10 %.0f, p90 %.0f) -> that many MHz if s_memtime counts shader cycles"
      % (tpu.mean(), np.percentile(tpu, 10), np.percentile(tpu, 90)))
# residency and workgroup life over the launch, in tenths of the span
edges = np.linspace(rt0.min(), rt1.max(), 11)
for i in range(10):
    a, b = edges[i], edges[i + 1]
    resident = (np.minimum(rt1, b) - np.maximum(rt0, a)).clip(min=0).sum() / (b - a)
    started = (rt0 >= a) & (rt0 < b)
    print("  %3d - %3d %% of the span: %6.1f workgroups resident (%.2f per CU), %5d started, mean life of those %.1f us"
          % (10 * i, 10 * i + 10, resident, resident / 256.0, started.sum(), life[started].mean() if started.any() else 0.0))
# phase durations (chip clock, 100 MHz has no phase stamps: scaled from s_memtime by each workgroup's own life) of the
# workgroups started in each fifth of the span
h2 = None
ph = np.diff(t[:, :6], axis=1).astype(np.float64)
scale = (life / ph.sum(axis=1))[:, None]          # us per tick, per workgroup
phus = ph * scale
for i in range(5):
    a, b = edges[2 * i], edges[2 * i + 2]
    sel = (rt0 >= a) & (rt0 < b) & (t[:, 6] == 0)
    if sel.any():
        m = phus[sel].mean(axis=0)
        print("  started in %3d - %3d %%: %5d rows without flags, life %.1f us = staging %.1f + moments %.1f + scores %.1f + flags %.1f + transforms %.1f"
              % (20 * i, 20 * i + 20, sel.sum(), m.sum(), m[0], m[1], m[2], m[3], m[4]))
h.close()
