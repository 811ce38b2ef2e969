"""Throughput of the downstream search (BASELINE config 5) on the GPU box: heimdall's production
settings (DM 2-1000, boxcars up to 64, 4096 channels, gulp 30720 samples = 24 s of filterbank),
dm_step 2 -> 500 trial DMs.  Prints the time per gulp, the stage times and the real-time factor for
(a) full S/N planes back to the host, (b) the peak list only, codes from page-locked host memory,
(c) the peak list only, codes already on the device (as they are behind process_baseband)."""
import importlib
import sys
import time
import os

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
search = importlib.import_module("vlite-fast_amd.search")

rng = np.random.default_rng(1)
T = search.HEIMDALL_GULP
pinned = torch.empty((T, 4096), dtype=torch.uint8, pin_memory=True)
codes = pinned.numpy()
codes[:] = np.clip(rng.normal(127.5, 1 / 0.02957, (T, 4096)), 0, 255).astype(np.uint8)
dcodes = pinned.cuda()
torch.cuda.synchronize()
for step in (2.0, 10.0):
    with search.Searcher(max_samples=T, dm_step=step) as s:
        s.set_baseline(2560)
        tout = T - s.max_delay
        for name, fn in (("planes to host", lambda: s.run(codes)),
                         ("peak list, host codes", lambda: s.peaks(codes, 6.0)),
                         ("peak list, device codes", lambda: s.peaks(None, 6.0, device_ptr=dcodes.data_ptr(), nsamp=T))):
            fn()
            t0 = time.perf_counter()
            n = 5
            for _ in range(n):
                r = fn()
            dt = (time.perf_counter() - t0) / n
            print("dm_step %g, %-24s %d DMs x %d boxcars, %d new samples per %d-sample gulp: %.2f ms per gulp = %.0fx real time; "
                  "stages (ms) %s" % (step, name + ":", s.ndm, s.nbox, tout, T, dt * 1e3, tout * s.tsamp / dt,
                                      {k: round(v, 2) for k, v in s.timers().items()}))
