"""Trigger wire format downstream of the search: candidate -> trigger_t -> multicast.

The reference chain: heimdall candidates -> src/trigger.py (coincidence, trigger criteria) ->
multicast 224.3.29.71:20003 -> src/dumper.c.  Two formats exist in the reference:
  * the C struct dumper expects, src/utils.h:47-57:
        struct { double t0, t1; float sn, dm, width, peak_time; char meta[128]; }   = 160 bytes
  * the older 144-byte 'dd128s' that src/trigger.py:174 and src/recorder.py:49 still pack and that
    dumper rejects by size (src/dumper.c:522-526).
This module writes the 160-byte form and reads both.
"""
import calendar
import socket
import struct
import time

TRIGGER_GROUP = ("224.3.29.71", 20003)              # src/multicast.h:14,19; src/trigger.py:75
TRIGGER_FMT = "=ddffff128s"
TRIGGER_SIZE = struct.calcsize(TRIGGER_FMT)         # 160
LEGACY_FMT = "=dd128s"                              # 144
DM_DELAY = 4.15e-3 * (0.320 ** -2 - 0.384 ** -2)    # s per DM unit across the band, src/trigger.py:33


def pack_trigger(t0, t1, sn, dm, width, peak_time, meta=""):
    m = meta.encode("ascii", "replace")[:127] if isinstance(meta, str) else bytes(meta)[:127]
    return struct.pack(TRIGGER_FMT, float(t0), float(t1), float(sn), float(dm), float(width), float(peak_time), m)


def unpack_trigger(buf):
    if len(buf) == TRIGGER_SIZE:
        t0, t1, sn, dm, width, peak, meta = struct.unpack(TRIGGER_FMT, buf)
        return dict(t0=t0, t1=t1, sn=sn, dm=dm, width=width, peak_time=peak, meta=meta.split(b"\0", 1)[0].decode("ascii", "replace"))
    if len(buf) == struct.calcsize(LEGACY_FMT):
        t0, t1, meta = struct.unpack(LEGACY_FMT, buf)
        return dict(t0=t0, t1=t1, sn=None, dm=None, width=None, peak_time=None, meta=meta.split(b"\0", 1)[0].decode("ascii", "replace"))
    raise ValueError("not a trigger_t: %d bytes" % len(buf))


def passes_criteria(cand, tsamp, nbeam=1, snthresh=8, minbeam=3, wmax=0.01, dmmin=70):
    """The reference's trigger test, src/trigger.py:45-66 (width in seconds)."""
    width = (cand["i1"] - cand["i0"]) * tsamp
    return nbeam >= minbeam and width < wmax and cand["dm"] > dmmin and cand["snr"] > snthresh


def trigger_for_candidate(cand, utc_start, tsamp):
    """Dump window of a candidate as src/trigger.py:154-174 computes it: from 0.1 s before i0 to
    0.1 s after i1 plus the dispersion sweep across the band."""
    dm_delay = cand["dm"] * DM_DELAY
    dump_offs = cand["i0"] * tsamp
    dump_len = (cand["i1"] - cand["i0"]) * tsamp + dm_delay
    t = time.strptime(utc_start, "%Y-%m-%d-%H:%M:%S")
    t0 = calendar.timegm(t) + dump_offs - 0.1
    t1 = t0 + dump_len + 0.2
    meta = "Trigger at UTC %s + %d" % (utc_start, dump_offs)
    return pack_trigger(t0, t1, cand["snr"], cand["dm"], (cand["i1"] - cand["i0"]) * tsamp, cand["peak_time"], meta)


def send_trigger(payload, group=TRIGGER_GROUP):
    sock = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    sock.settimeout(0.2)
    sock.setsockopt(socket.IPPROTO_IP, socket.IP_MULTICAST_TTL, struct.pack("b", 1))
    try:
        sock.sendto(payload, group)
    finally:
        sock.close()
