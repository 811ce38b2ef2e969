"""The oracle's K-chain, frozen: oracle.segment (oracle/pb_oracle.c, the restatement of src/pb_kernels.cu K1..K11 run
in the order of src/process_baseband.cu:1108-1376) must still produce, bit for bit, what tests/golden/oracle_kchain.json
recorded for the same seeded inputs -- codes of both streams, the fp32 planes, row weights, D'Agostino scores with
their kurtosis and power, and the bandpass state -- over R in {16, 64} x RFI modes 0/1/2 x 8/4/2 bits x npol 1/2, an
injected-FRB case and one production-size (R = 1024) segment.

The HIP kernels are held to the oracle bit for bit by the GPU tests; K2..K5 and K7..K11 have no reference-held vector.
This fixture is what keeps oracle and kernels from drifting TOGETHER: a change to pb_oracle.c's arithmetic fails here
unless the fixture is regenerated (python tests/golden/make_kchain_golden.py) in the same commit, with the reason in
its message."""
import json
import os

import pytest

import make_kchain_golden as G

FIX = json.load(open(G.OUT))


def test_fixture_covers_the_matrix():
    names = set(FIX["cases"])
    for R in (16, 64):
        for rfi in (0, 1, 2):
            for nbit in (8, 4, 2):
                for npol in (1, 2):
                    assert "R%d_r%d_b%d_P%d" % (R, rfi, nbit, npol) in names
    assert "R1024_r2_b8_P1" in names and "R64_r2_b8_P1_frb" in names
    assert [c["name"] for c in G.cases()] == sorted(names, key=[c["name"] for c in G.cases()].index)
    # the inputs exercise what they claim to: flags in the modes that flag, different streams in mode 2
    d = FIX["cases"]["R64_r2_b8_P1"]["digests"]
    assert d["peek"]["nflagged_last_seg"] > 10 and d["codes_raw"] != d["codes_kur"] and d["bp_raw"] != d["bp_kur"]
    assert FIX["cases"]["R64_r2_b8_P1_frb"]["digests"]["codes_raw"] != d["codes_raw"]


@pytest.mark.parametrize("group", ["R16", "R64", "R1024"])
def test_oracle_kchain_matches_frozen_digests(group):
    got = G.compute(select=lambda c: c["name"].startswith(group + "_"))
    assert got, group
    for name, rec in got.items():
        want = FIX["cases"][name]["digests"]
        assert rec["digests"]["input"] == want["input"], "%s: the seeded INPUT changed (tests/helpers.make_input)" % name
        for k, v in rec["digests"].items():
            assert v == want[k], "%s: oracle output `%s` differs from the frozen fixture" % (name, k)
