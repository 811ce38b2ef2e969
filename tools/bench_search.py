"""Throughput of the downstream search (BASELINE config 5) on the GPU box: heimdall's production
settings (DM 2-1000, boxcars up to 64, 4096 channels, gulp 30720 samples = 24 s of filterbank),
dm_step 2 -> 500 trial DMs.  Prints the time per gulp and the real-time factor."""
import importlib
import sys
import time
import os

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
search = importlib.import_module("vlite-fast_amd.search")

rng = np.random.default_rng(1)
T = search.HEIMDALL_GULP
codes = np.clip(rng.normal(127.5, 1 / 0.02957, (T, 4096)), 0, 255).astype(np.uint8)
for step in (2.0, 10.0):
    with search.Searcher(max_samples=T, dm_step=step) as s:
        s.run(codes)
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            r = s.run(codes)
        dt = (time.perf_counter() - t0) / n
        tout = T - s.max_delay
        print("dm_step %g: %d DMs x %d boxcars, %d samples out of a %d-sample gulp: %.1f ms per gulp "
              "(%.0fx real time for %.1f s of new data), incl. H2D of the codes and D2H of snr/width planes"
              % (step, s.ndm, s.nbox, tout, T, dt * 1e3, tout * s.tsamp / dt, tout * s.tsamp))
