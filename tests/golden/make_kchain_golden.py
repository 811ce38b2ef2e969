#!/usr/bin/env python3
"""Freeze what the C oracle (oracle/pb_oracle.c: the restatement of src/pb_kernels.cu K1..K11 in the order of
src/process_baseband.cu:1108-1376) produces for seeded inputs: tests/golden/oracle_kchain.json holds sha256 digests
of every output of oracle.segment -- raw and excised codes, the fp32 pre-quantisation planes, the row weights, the
D'Agostino scores with the kurtosis and power they come from, and the bandpass state after the last segment -- plus a
few hundred bytes / floats in clear so that a mismatch can be located.

Why: K2..K5 and K7..K11 have no reference-held vector (pb_kernels.cu cannot be built here), so the oracle is the only
statement of those kernels; the HIP kernels are tested against it bit for bit, and nothing else would notice if oracle
and kernels drifted TOGETHER.  With this fixture any change to pb_oracle.c's arithmetic turns
tests/test_oracle_kchain.py red; regenerate (python tests/golden/make_kchain_golden.py) only in the same commit as the
change, with the reason in the commit message.

Inputs: tests/helpers.make_input (splitmix64 integer hashing -> the same bytes on any NumPy).  Cases: R in {16, 64} x
RFI modes 0/1/2 x nbit 8/4/2 x npol 1/2 (3 segments each, RFI bursts, a fully flagged row, a dropped frame), an
injected-FRB case, and ONE production-size segment (R = 1024)."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

OUT = os.path.join(HERE, "oracle_kchain.json")


def cases():
    out = []
    for R in (16, 64):
        for rfi_mode in (0, 1, 2):
            for nbit in (8, 4, 2):
                for npol in (1, 2):
                    out.append(dict(name="R%d_r%d_b%d_P%d" % (R, rfi_mode, nbit, npol), R=R, nseg=3, seed=500 + R, rfi_mode=rfi_mode,
                                    nbit=nbit, npol=npol, frb=0))
    out.append(dict(name="R64_r2_b8_P1_frb", R=64, nseg=3, seed=577, rfi_mode=2, nbit=8, npol=1, frb=1))
    out.append(dict(name="R1024_r2_b8_P1", R=1024, nseg=1, seed=42, rfi_mode=2, nbit=8, npol=1, frb=0))
    return out


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_case(O, c):
    from helpers import make_input
    data = make_input(c["seed"], c["R"], c["nseg"])
    bp_raw = np.zeros(2 * O.NCHAN, np.float32)
    bp_kur = np.zeros(2 * O.NCHAN, np.float32)
    delays = O.set_frb_delays(30.0, c["R"]) if c["frb"] else None
    keys = ("codes_raw", "codes_kur", "ave_raw", "ave_kur", "weights", "dag", "pow", "kur")
    h = {k: hashlib.sha256() for k in keys}
    inj = 1 if c["frb"] else 0
    last = None
    for s in range(c["nseg"]):
        r = O.segment(data[s], c["R"], bp_raw, bp_kur, rfi_mode=c["rfi_mode"], npol=c["npol"], nbit=c["nbit"],
                      frb_delays=delays, inject_now=inj)
        if inj:
            inj += 1
        for k in keys:
            h[k].update(np.ascontiguousarray(getattr(r, k)).tobytes())
        last = r
    rec = {k: h[k].hexdigest() for k in keys}
    rec["bp_raw"], rec["bp_kur"], rec["input"] = sha(bp_raw), sha(bp_kur), sha(data)
    # a little in clear: the first bytes of the last segment's codes, the last weights, the flag count
    rec["peek"] = {"codes_kur_hex": last.codes_kur[:32].tobytes().hex(), "codes_raw_hex": last.codes_raw[:32].tobytes().hex(),
                   "weights_tail_hex": last.weights[-8:].tobytes().hex(),
                   "bp_kur_head_hex": bp_kur[O.CHANMIN:O.CHANMIN + 8].tobytes().hex(),
                   "nflagged_last_seg": int((last.dag > 3.0).sum()) if c["rfi_mode"] else 0}
    return rec


def compute(select=None):
    import oracle as O
    O.lib()
    out = {}
    for c in cases():
        if select is not None and not select(c):
            continue
        out[c["name"]] = dict(case=c, digests=run_case(O, c))
    return out


def main():
    got = compute()
    with open(OUT, "w") as f:
        json.dump({"generator": "tests/golden/make_kchain_golden.py", "oracle": "oracle/pb_oracle.c (strict build)",
                   "cases": got}, f, indent=1, sort_keys=True)
    print("wrote %s: %d cases" % (OUT, len(got)))


if __name__ == "__main__":
    main()
