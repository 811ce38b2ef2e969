"""Host-side formats on the CPU: VDIF frames, ASCII ring headers, SIGPROC headers, file names,
the genbase-style generator.  Pinned to the reference's own Python decoder where it has one
(tests/golden: VDIFHeader, get_data)."""
import importlib
import os
import struct

import numpy as np
import pytest

import synth

vdif = importlib.import_module("vlite-fast_amd.vdif")
sigproc = importlib.import_module("vlite-fast_amd.sigproc")
dada = importlib.import_module("vlite-fast_amd.dada")
genbase = importlib.import_module("vlite-fast_amd.genbase")


def test_vdif_header_pack_matches_reference_decoder(golden):
    for w, f in zip(golden["h8_words"], golden["h8_fields"]):
        sec, ep, fr, flen, nsamp, st, thid, th = [int(x) for x in f]
        mine = vdif.pack_header(sec, ep, fr, st, thid)
        assert np.array_equal(mine, w)
        d = vdif.unpack_header(mine.tobytes())
        assert (d["second"], d["epoch"], d["frame"], d["frame_length"], d["frame_nsamp"], d["station"],
                d["threadid"], d["thread"]) == (sec, ep, fr, flen, nsamp, st, thid, th)
    h = dict(epoch=33, second=3600, frame=0)
    assert vdif.vdif_to_unixepoch(h) == int(golden["h8_unix_ep33_sec3600"])
    # 2016-07-01 01:00:00 UTC = MJD 57570 + 3600 s
    assert vdif.frame_mjd(h) == 57570 and vdif.frame_mjd_sec(h) == 3600
    assert abs(vdif.frame_dmjd(dict(epoch=33, second=3600, frame=12800)) - (57570 + 3600.5 / 86400)) < 1e-12
    ep, sec = vdif.epoch_for_unix(int(golden["h8_unix_ep33_sec3600"]))
    assert (ep, sec) == (33, 3600)


def test_framer_roundtrip_and_reference_get_data(golden, oracle):
    n = 8 * 5000
    p0 = synth.baseband_u8(1, n)
    p1 = synth.baseband_u8(2, n)
    blk = vdif.frame_block(p0, p1, second=100, epoch=33, station=5, frame0=25596)
    assert blk.size == 16 * 5032
    h0 = vdif.unpack_header(blk[:32].tobytes())
    h_last = vdif.unpack_header(blk[-5032:-5000].tobytes())
    assert (h0["second"], h0["frame"], h0["thread"], h0["station"], h0["nbit"]) == (100, 25596, 0, 5, 8)
    assert (h_last["second"], h_last["frame"], h_last["thread"]) == (101, 3, 1)      # second rolled over
    d = vdif.deframe_block(blk)
    assert np.array_equal(d[0], p0) and np.array_equal(d[1], p1)
    # the reference's own reader (restated in the oracle, pinned by tests/golden) agrees
    ref = oracle.vdif_get_data(blk)
    assert np.array_equal(ref[0], p0.astype(np.float32) - np.float32(127.5))
    assert np.array_equal(ref[1], p1.astype(np.float32) - np.float32(127.5))
    # out-of-order arrival and a dropped frame (zero fill, src/writer.c:674-688)
    fr = blk.reshape(16, 5032).copy()
    perm = np.array([1, 0, 3, 2, 5, 4, 7, 6, 9, 8, 11, 10, 13, 12, 15, 14])
    shuffled = fr[perm]
    shuffled[6, 0:4] = np.frombuffer(struct.pack("<I", 0x80000000 | 100), np.uint8)   # invalid bit
    d2 = vdif.deframe_block(np.concatenate([fr[:1].ravel(), shuffled[1:].ravel()]))
    assert np.array_equal(d2[0, :5000], p0[:5000])


def test_ascii_header_roundtrip():
    h = vdif.writer_header(station=12, ra=0.8718, dec=-0.72452, name="B0833-45", scanstart=58000.5,
                           dataid="19A-331.sb1", epoch=33, second=3600)
    raw = vdif.ascii_header_format(h)
    assert len(raw) == 4096
    p = vdif.ascii_header_parse(raw)
    assert p["STATIONID"] == "12" and p["NAME"] == "B0833-45" and float(p["DEC"]) == -0.72452
    assert p["UTC_START"] == "2016-07-01-01:00:00" and p["NBIT"] == "8"


def test_sigproc_header_bytes():
    hdr = sigproc.sigproc_header(station_id=7, ra=0.87180, dec=-0.72452, name="B0833-45",
                                 dmjd=57570.5, npol=1, nbit=8)
    d, n = sigproc.read_header(hdr)
    assert n == len(hdr)
    assert hdr.startswith(struct.pack("=i", 12) + b"HEADER_START" + struct.pack("=i", 11) + b"source_name")
    assert list(d.keys()) == ["source_name", "barycentric", "telescope_id", "src_raj", "src_dej", "data_type",
                              "fch1", "foff", "nchans", "nbits", "tstart", "tsamp", "nifs"]
    assert d["source_name"] == "B0833-45" and d["telescope_id"] == 7 and d["nchans"] == 4096
    assert d["fch1"] == 384 + (2155 - 0.5) * (-64. / 6251) and abs(d["fch1"] - 361.941449) < 1e-5
    assert d["foff"] == -64. / 6251 and d["tsamp"] == 12500. / 128000000 * 8 == 0.00078125
    assert d["nbits"] == 8 and d["nifs"] == 1 and d["tstart"] == 57570.5 and d["data_type"] == 1
    # RA 0.8718 rad = 3h19m48.1s, computed in C float; the declination's sign is dropped
    assert abs(d["src_raj"] - 31948.1) < 0.1
    assert d["src_dej"] > 0 and abs(d["src_dej"] - 413042.97) < 0.1
    f = np.float32
    hh = f((180 / np.pi) * (24. / 360) * 0.87180)
    mm = f(f(hh - f(int(hh))) * f(60))
    ss = f(f(mm - f(int(mm))) * f(60))
    assert d["src_raj"] == float(f(int(hh) * 1e4 + int(mm) * 1e2 + float(ss)))


def test_file_names():
    t = 1467334800   # 2016-07-01 01:00:00 UTC
    fb, fbk, co, cok = sigproc.fb_names(t, 7, "/mnt/ssd/fildata")
    assert fb == "/mnt/ssd/fildata/20160701_010000_muos_ea07.fil"
    assert fbk == "/mnt/ssd/fildata/20160701_010000_muos_ea07_kur.fil"
    assert co == "/mnt/ssd/fildata/20160701_010000_muos_ea99.fil" and cok.endswith("_muos_ea99_kur.fil")
    assert sigproc.change_extension("abc.fil", ".fil", ".histo") == "abc.histo"
    assert sigproc.change_extension("abc", ".fil", ".histo") == "abc.histo"


def test_process_baseband_cli_matches_reference_getopt():
    pb = importlib.import_module("vlite-fast_amd.process_baseband")
    # scripts/start_process:50 and scripts/baseband_test:26
    a = pb.build_parser().parse_args("-k 40 -K 42 -w 2 -b 2 -g 1 -o -C 46".split())
    assert (a.key_in, a.key_out, a.key_co, a.write_fb, a.nbit, a.gpu_id, a.stdout_output) == (0x40, 0x42, 0x46, 2, 2, 1, True)
    assert (a.rfi_mode, a.npol) == (2, 1)                       # reference defaults
    b = pb.build_parser().parse_args("-p 0 -k 40 -K 0 -w 1 -b 2 -P 1 -r 2 -o".split())
    assert b.legacy_p == "0" and b.key_out == 0 and b.write_fb == 1
    for bad in ("-b 3", "-r 5", "-P 4"):
        with pytest.raises(SystemExit):
            pb.validate(pb.build_parser().parse_args(bad.split()))
    assert pb.source_allowed({"NAME": "B0329+54"}) and not pb.source_allowed({"NAME": "3C286"})
    assert pb.source_allowed({"NAME": "X", "DATAID": "19A-331.sb3"})


def test_memory_and_file_rings(tmp_path):
    r = dada.MemoryRing()
    r.write_header(b"A 1\n".ljust(4096, b"\0"))
    r.write(np.arange(10, dtype=np.uint8))
    r.end_of_data()
    assert r.next_header().startswith(b"A 1") and r.read(4) == bytes([0, 1, 2, 3]) and len(r.read(100)) == 6
    r.finish_observation()
    assert r.next_header() is None
    p = tmp_path / "d.uw"
    dada.write_dump(str(p), b"K v\n".ljust(4096, b"\0"), np.arange(7, dtype=np.uint8))
    fr = dada.FileRing(str(p))
    assert vdif.ascii_header_parse(fr.next_header()) == {"K": "v"}
    assert fr.read(5) == bytes(range(5)) and fr.read(5) == bytes([5, 6]) and fr.next_header() is None
    with pytest.raises(RuntimeError, match="psrdada"):
        dada.open_ring(0x40)


def test_file_ring_over_a_fifo(tmp_path):
    """A FIFO fed in small pieces (how a psrdada bridge process would deliver a live ring): requests are
    still filled completely, readinto fills a staging array in place, EOF gives a short read."""
    import os
    import threading
    path = str(tmp_path / "ring.fifo")
    os.mkfifo(path)
    hdr = b"STATIONID 7\n".ljust(4096, b"\0")
    payload = (np.arange(100000, dtype=np.uint32) % 251).astype(np.uint8).tobytes()

    def feed():
        with open(path, "wb", buffering=0) as f:
            data = hdr + payload
            for i in range(0, len(data), 3001):       # odd-sized pieces
                f.write(data[i:i + 3001])

    t = threading.Thread(target=feed, daemon=True)
    t.start()
    fr = dada.FileRing(path)
    assert vdif.ascii_header_parse(fr.next_header()) == {"STATIONID": "7"}
    a = fr.read(50000)
    assert a == payload[:50000]
    buf = np.zeros(40000, np.uint8)
    assert fr.readinto(buf) == 40000 and buf.tobytes() == payload[50000:90000]
    tail = fr.read(50000)
    assert tail == payload[90000:]                    # short read at end of data
    assert fr.read(10) == b""
    t.join(timeout=10)
    assert not t.is_alive()


def test_genbase_recipe_small():
    assert genbase.dm_samples(30.0) == (1141760, 1526784) or sum(genbase.dm_samples(30.0)) > 2e6
    n_lo, n_hi = genbase.dm_samples(30.0)
    # DM 30: 95.6 ms smear across the band at 128 MS/s
    assert abs((n_lo + n_hi) / 128e6 - 30 / 2.41e-10 * (320.**-2 - 384.**-2) * 1e-6) < 2e-5
    # a short run at a reduced rate: pulse every 0.02 s, DM small enough for a 2^18 buffer
    rate = 1280000
    chunks = list(genbase.generate(tobs=1.0, dm=0.002, period=0.02, ampl=1.0, seed=3, buflen=1 << 18, rate=rate))
    p0 = np.concatenate([c[0] for c in chunks])
    p1 = np.concatenate([c[1] for c in chunks])
    assert p0.dtype == np.uint8 and p0.size == p1.size and p0.size >= rate
    x = p0.astype(np.float64) - 128.5
    assert abs(x.mean()) < 1.0 and 5 < x.std() < 20          # band-pass taper lowers sigma below 16.9
    per = int(0.02 * rate)
    nfold = x.size // per
    prof = (x[:nfold * per] ** 2).reshape(nfold, per).mean(axis=0)
    prof = prof.reshape(100, -1).mean(axis=1)
    assert prof.max() > 1.5 * np.median(prof)                # the pulse survives chirp + digitiser
    assert np.array_equal(genbase.digitize(np.array([-100., 0., 0.02957 * 2, 100.], np.float32)),
                          np.array([0, 128, 129, 255], np.uint8))


def test_genbase_writes_replayable_dump(tmp_path):
    rate = 1280000
    path = str(tmp_path / "sim.uw")
    nfr = genbase.write_observation(path, genbase.generate(tobs=1.2, dm=0.0, period=0.05, ampl=0.5, seed=4,
                                                            buflen=1 << 17, rate=rate), t_unix=1467334800,
                                    station=3, frames_per_sec=200)
    ring = dada.FileRing(path)
    hdr = vdif.ascii_header_parse(ring.next_header())
    assert hdr["NAME"] == "B0833-45" and hdr["STATIONID"] == "3" and hdr["UTC_START"] == "2016-07-01-01:00:00"
    data = np.frombuffer(ring.read(1 << 30), np.uint8)
    assert data.size == nfr * 2 * 5032
    h0 = vdif.unpack_header(data[:32].tobytes())
    assert (h0["epoch"], h0["second"], h0["frame"], h0["thread"]) == (33, 3600, 0, 0)
    h_roll = vdif.unpack_header(data[400 * 5032:400 * 5032 + 32].tobytes())
    assert (h_roll["second"], h_roll["frame"]) == (3601, 0)


def test_out_ring_header_all_keys_in_reference_order():
    """write_psrdada_header (src/process_baseband.cu:136-201): seventeen keys, this order, these printf
    formats (%d / %lf / %s / %lu), values from the incoming ring header and the first VDIF frame."""
    inhdr = vdif.writer_header(7, 0.8718, -0.72452, "B0833-45", 58000.25, "19A-331", 33, 3600)
    vh = vdif.unpack_header(vdif.pack_header(3600, 33, 0, 7, 0).tobytes())
    t_unix = vdif.vdif_to_unixepoch(vh)
    h = sigproc.psrdada_out_header(inhdr, vh, 1, 2, "/mnt/ssd/fildata/20160701_010000_muos_ea07_kur.fil", t_unix,
                                   vdif.frame_mjd(vh), vdif.frame_mjd_sec(vh))
    assert list(h.keys()) == ["STATIONID", "BEAM", "RA", "DEC", "NAME", "SCANSTART", "NCHAN", "BANDWIDTH", "CFREQ",
                              "NPOL", "NBIT", "TSAMP", "UTC_START", "UNIXEPOCH", "VDIF_MJD", "VDIF_SEC", "SIGPROC_FILE"]
    chbw = -64. / 6251
    expect = {"STATIONID": "7", "BEAM": "7", "RA": "0.871800", "DEC": "-0.724520", "NAME": "B0833-45",
              "SCANSTART": "58000.250000", "NCHAN": "4096", "BANDWIDTH": "%f" % (4096 * chbw),
              "CFREQ": "%f" % (384. + 0.5 * (2155 + 6250 - 1) * chbw), "NPOL": "1", "NBIT": "2",
              "TSAMP": "%f" % (12500. / 128000000 * 8 * 1e6), "UTC_START": "2016-07-01-01:00:00",
              "UNIXEPOCH": "1467334800.000000", "VDIF_MJD": "57570", "VDIF_SEC": "3600",
              "SIGPROC_FILE": "/mnt/ssd/fildata/20160701_010000_muos_ea07_kur.fil"}
    assert h == expect
    assert expect["TSAMP"] == "781.250000" and expect["BANDWIDTH"] == "-41.936330" and expect["CFREQ"] == "340.978403"
    # on the ring it is a 4096-byte block of "KEY value" lines that parses back to the same pairs
    raw = vdif.ascii_header_format(h)
    assert len(raw) == 4096 and vdif.ascii_header_parse(raw) == expect
