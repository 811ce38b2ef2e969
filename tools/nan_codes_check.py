"""How the 8/4/2-bit quantiser treats NaN planes (whole segments of dropped frames): GPU vs oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from helpers import compact_ave, libpb, make_input, oracle_run
import importlib
O = importlib.import_module("oracle.oracle")
lp = libpb()
R, NSEG = 16, 3
d = make_input(11, R, NSEG, rfi=False, dropped=False)
d[0] = 0
for nbit in (8, 4, 2):
    h = lp.PbHandle(nant=1, nbit=nbit, npol=1, rfi_mode=2, fft_backend=lp.FFT_LDS, rows_per_seg=R, max_seg=NSEG, keep_ave=True)
    for s in range(NSEG):
        h.submit_planar(0, s, d[s, 0], d[s, 1])
    h.process(NSEG)
    out = h.fetch(0, 0, NSEG, weights=True, ave=True)
    h.close()
    res, _, _ = oracle_run(O, d, R, rfi_mode=2, npol=1, nbit=nbit)
    for name in ("raw", "kur"):
        refa = np.concatenate([compact_ave(getattr(r, "ave_" + name), R, 1) for r in res])
        ref = np.concatenate([getattr(r, "codes_" + name) for r in res])
        nn = int(np.isnan(refa).sum())
        print(nbit, name, "NaN samples:", nn, "codes equal everywhere:", bool(np.array_equal(out[name], ref)),
              "gpu codes at NaN:", np.unique(out[name][: max(1, nn * nbit // 8)])[:4], "oracle:", np.unique(ref[: max(1, nn * nbit // 8)])[:4])
