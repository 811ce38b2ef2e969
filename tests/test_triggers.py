"""Trigger wire format (CPU): the 160-byte trigger_t of src/utils.h:47-57 and the window arithmetic
of src/trigger.py:154-174."""
import importlib
import struct

import pytest

tr = importlib.import_module("vlite-fast_amd.triggers")


def test_trigger_struct_layout():
    assert tr.TRIGGER_SIZE == 160
    p = tr.pack_trigger(1467334800.25, 1467334803.5, 12.5, 302.0, 0.003125, 2.34, "Trigger at UTC x + 2")
    assert len(p) == 160
    # field offsets of the C struct: two doubles, four floats, char[128]
    assert struct.unpack_from("=d", p, 0)[0] == 1467334800.25 and struct.unpack_from("=d", p, 8)[0] == 1467334803.5
    f4 = struct.unpack_from("=4f", p, 16)
    assert f4[:2] == (12.5, 302.0) and abs(f4[2] - 0.003125) < 1e-9 and abs(f4[3] - 2.34) < 1e-6
    assert p[32:32 + 20] == b"Trigger at UTC x + 2" and p[-1:] == b"\0"
    d = tr.unpack_trigger(p)
    assert d["dm"] == 302.0 and d["meta"] == "Trigger at UTC x + 2"
    legacy = struct.pack("=dd128s", 1.0, 2.0, b"old")
    assert tr.unpack_trigger(legacy)["sn"] is None and tr.unpack_trigger(legacy)["meta"] == "old"
    with pytest.raises(ValueError):
        tr.unpack_trigger(b"x" * 10)


def test_trigger_window_follows_reference():
    cand = dict(snr=20.0, dm=100.0, i0=1280, i1=1284, peak_time=1.0, peak_idx=1281)
    tsamp = 1. / 1280
    p = tr.trigger_for_candidate(cand, "2016-07-01-01:00:00", tsamp)
    d = tr.unpack_trigger(p)
    dm_delay = 100.0 * 4.15e-3 * (0.320 ** -2 - 0.384 ** -2)
    assert abs(d["t0"] - (1467334800 + 1.0 - 0.1)) < 1e-6
    assert abs((d["t1"] - d["t0"]) - (4 * tsamp + dm_delay + 0.2)) < 1e-6
    assert d["meta"] == "Trigger at UTC 2016-07-01-01:00:00 + 1"
    assert tr.passes_criteria(cand, tsamp, nbeam=3) and not tr.passes_criteria(cand, tsamp, nbeam=1)
    assert not tr.passes_criteria(dict(cand, dm=50.0), tsamp, nbeam=3)
