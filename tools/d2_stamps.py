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
PIPE = os.environ.get("D2_PIPE") == "1"      # as benchmarked: three buffer sets, detect beside the next channeliser
NSETS = 3 if PIPE else 1
h = lp.PbHandle(device=0, nant=1, nbit=8, npol=1, rfi_mode=2, rows_per_seg=1024, max_seg=S, nsets=NSETS,
                taps=int(os.environ.get("D2_TAPS", "1")))
sec = synth_second(torch, dev, 42, h.seg_samples, S)
torch.cuda.synchronize()
for st in range(NSETS):
    h.select_set(st)
    for s in range(S):
        h.submit_planar_dev(0, s, sec[s][0].data_ptr(), sec[s][1].data_ptr(), h.seg_samples)
h.sync()
def stamps():
    o = (C.c_ulonglong * 32)()
    rc = L.pb_internal_d2_stamps(o)
    assert rc == 0
    return list(o)


def run(nb):
    for k in range(nb):
        h.select_set(k % NSETS)
        h.process(S)
        if not PIPE:
            h.sync()
        elif k >= 2:
            h.select_set((k - 2) % NSETS)
            h.fetch_view(0, 1, S)
    h.sync()


run(12)
before = stamps()
run(42 if PIPE else 5)     # pipelined: 40 of the 42 detects run beside the next batch's channeliser
after = stamps()
out = [a - b for a, b in zip(after, before)]
rc = 0
nl = max(1, out[3])
names = {0: "A", int(os.environ.get("D2_WAVE_L", "5")): "L", 1: "B0", 2: "B1", 3: "B2", 9 - int(os.environ.get("D2_WAVE_L", "5")): "B3"}
nstep = 10 * 1024 // 32 + 2
print("rc", rc, "steps", nstep, "launches", nl, "pipelined" if PIPE else "alone")
for w in range(6):
    work, wait, extra = out[w * 4] / nl, out[w * 4 + 1] / nl, out[w * 4 + 2] / nl
    print("wave %d %-3s work %8d (%6.0f/step)  barrier wait %8d (%6.0f/step)  dma wait %8d (%6.0f/step)"
          % (w, names.get(w, "?"), work, work / nstep, wait, wait / nstep, extra, extra / nstep))
h.close()

# when every workgroup of a launch started and ended (last 64 launches)
if hasattr(L, "pb_internal_d2_wg") or True:
    import numpy as np
    buf = (C.c_ulonglong * (64 * 256 * 4))()
    if L.pb_internal_d2_wg(buf) == 0:
        a = np.frombuffer(buf, dtype=np.uint64).reshape(64, 256, 4).astype(np.int64)
        n = a[:, 0, 3]
        order = np.argsort(n)
        for i in order[-8:-2]:
            t0 = a[i, :, 0].min()
            st = (a[i, :, 0] - t0) / 100.0          # us
            en = (a[i, :, 1] - t0) / 100.0
            print("launch %3d: workgroup starts (us) min %.1f median %.1f p90 %.1f max %.1f | life median %.1f max %.1f | last end %.1f"
                  % (n[i], st.min(), np.median(st), np.percentile(st, 90), st.max(), np.median(en - st), (en - st).max(), en.max()))
