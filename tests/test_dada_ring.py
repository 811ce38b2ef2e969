"""psrdada ring endpoints: vlite-fast_amd/dada.PsrdadaRing over the flat shim of include/pb_dada.h
-- the REAL shim, vlite-fast_amd/csrc/pb_dada_shim.c, compiled by the package's `make dada` (-Wall -Wextra -Werror)
and executed.  psrdada is absent here, so UNDER the shim sits tests/mock_psrdada: declarations of exactly the psrdada
calls the shim makes, written from the reference's call sites, over rings that are shared files (two processes can
use them).  That stand-in pins nothing about psrdada's ABI; what is under test is the shim's own logic (locking,
header hand-over, block-level reads across buffer boundaries, the empty end-of-data buffer, refusal to mix the two
read interfaces, release on close), the ctypes binding, the call order, and that `process_baseband -k/-K/-C` produces
the reference's ring traffic: header + frames in; header, one write per segment (coadd ring); header, 10 s then 1 s
(heimdall ring).  The shim also runs under AddressSanitizer + UBSan and under ThreadSanitizer
(tests/mock_psrdada/shim_driver.c).  Reference: src/process_baseband.cu:541-569,799-838,1416-1422,1482-1513."""
import ctypes as C
import importlib
import os
import re
import subprocess

import numpy as np
import pytest

from test_host_loop import FPS, SEG, TRIM, FakeHandle, _args, _expected, _frames, _header

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dada = importlib.import_module("vlite-fast_amd.dada")
vdif = importlib.import_module("vlite-fast_amd.vdif")
pbmod = importlib.import_module("vlite-fast_amd.process_baseband")


@pytest.fixture()
def mock(psrdada_mock):
    return psrdada_mock


def _decls():
    txt = open(os.path.join(ROOT, "include", "pb_dada.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pb_dada_[a-z_]+)\s*\(", txt)))


def test_shim_and_mock_define_every_declared_symbol(mock):
    names = _decls()
    assert names == ["pb_dada_close", "pb_dada_end_read", "pb_dada_end_write", "pb_dada_next_header", "pb_dada_open",
                     "pb_dada_read", "pb_dada_read_mt", "pb_dada_write", "pb_dada_write_header"]
    shim = open(os.path.join(ROOT, "vlite-fast_amd", "csrc", "pb_dada_shim.c")).read()
    exported = subprocess.run(["nm", "-D", "--defined-only", mock.shim_path], stdout=subprocess.PIPE, check=True).stdout.decode()
    for n in names:
        assert hasattr(mock.shim, n) and re.search(r" T %s\b" % n, exported), n
        assert re.search(r"\b%s\s*\(pb_dada \*d|\b%s\s*\(uint32_t key" % (n, n), shim), n
    # the shim is nothing but the reference's psrdada calls (+ psrdada's block-level pair for pb_dada_read_mt)
    for call in ("ipcio_open_block_read", "ipcio_close_block_read","dada_hdu_create", "dada_hdu_set_key", "dada_hdu_connect", "dada_hdu_lock_read", "ipcbuf_get_next_read",
                 "ipcbuf_mark_cleared", "ipcio_read", "dada_hdu_unlock_read", "dada_hdu_lock_write",
                 "ipcbuf_get_next_write", "ipcbuf_mark_filled", "ipcio_write", "dada_hdu_unlock_write"):
        assert call in shim


def test_open_ring_without_the_shim_says_how_to_build_it(monkeypatch):
    monkeypatch.setattr(dada, "SHIM_PATH", "/nonexistent/libpb_dada.so")
    monkeypatch.setattr(dada, "_SHIM", None)
    with pytest.raises(RuntimeError, match="make -C vlite-fast_amd/csrc dada"):
        dada.open_ring(0x40)


def test_ring_roundtrip_and_protocol(mock):
    assert mock.ctl.mock_psrdada_create(0x77, 4096, 8) == 0
    try:
        with pytest.raises(RuntimeError, match="key=78"):
            dada.PsrdadaRing(0x78, "r", lib=mock.shim)
        w = dada.PsrdadaRing(0x77, "w", lib=mock.shim)
        with pytest.raises(IOError):
            w.write(np.zeros(8, np.uint8))                      # data before a header
        w.write_header(_header())
        w.write(np.arange(10000, dtype=np.uint8))
        w.end_of_data()
        r = dada.PsrdadaRing(0x77, "r", lib=mock.shim)
        assert r.next_header() == bytes(_header())
        assert r.read(6000) == (np.arange(6000) % 256).astype(np.uint8).tobytes()
        buf = np.zeros(8000, np.uint8)
        assert r.readinto(buf) == 4000 and np.array_equal(buf[:4000], (np.arange(6000, 10000) % 256).astype(np.uint8))
        assert r.read(100) == b""                                # end of data
        r.finish_observation()
        assert mock.counts(0x77) == (3, 3, 1, 1)                 # 2 full buffers + the end-of-data one, all handed back
        mock.ctl.mock_psrdada_shutdown(0x77)
        assert r.next_header() is None                           # ring shut down
        r.close()
        w.close()
    finally:
        mock.ctl.mock_psrdada_destroy(0x77)


def test_block_level_reads_equal_ipcio_reads_and_never_mix(mock, monkeypatch):
    """pb_dada_read_mt (the ring's filled buffers, handed back when fully consumed, big pieces copied by several
    threads) returns the same byte stream as pb_dada_read for any request sizes -- frame-sized probes, seconds
    that straddle buffers, the short last buffer -- and the two are not mixed within an observation."""
    assert mock.ctl.mock_psrdada_create(0x51, 3 * 5032 * 100, 10) == 0    # 1.5 MB buffers
    try:
        stream = (np.arange(10 * 5032 * 100 + 777, dtype=np.uint64) * 2654435761 >> 7).astype(np.uint8)
        w = dada.PsrdadaRing(0x51, "w", lib=mock.shim)
        for _ in range(2):
            w.write_header(_header())
            w.write(stream)
            w.end_of_data()
        # observation 1: block level (default PB_DADA_THREADS = 8)
        r = dada.PsrdadaRing(0x51, "r", lib=mock.shim)
        assert r._threads == 8 and r.next_header() is not None
        got = [r.read(5032)]                                         # the host's first-frame probe
        big = np.zeros(4 * 5032 * 100, np.uint8)
        while True:
            n = r.readinto(big)
            got.append(big[:n].tobytes())
            if n < big.size:
                break
        assert r.read(10) == b""
        assert b"".join(got) == stream.tobytes()
        assert mock.counts(0x51)[:2] == (8, 4)                       # 3 full buffers + the short one, all handed back
        with pytest.raises(IOError):
            r._how = 1
            r.read(8)                                                # ipcio_read after block-level reads: refused
        r._how = 2
        r.finish_observation()
        # observation 2: the reference's ipcio_read only
        monkeypatch.setenv("PB_DADA_THREADS", "1")
        r2 = dada.PsrdadaRing(0x51, "r", lib=mock.shim)
        r2.close()
        r2._d, r._d = r._d, None                                     # (the first reader's handle: the ring has one read position)
        assert r2._threads == 1 and r2.next_header() is not None
        assert r2.read(5032) + r2.read(stream.size) == stream.tobytes()
        r2.finish_observation()
        assert mock.counts(0x51) == (8, 8, 2, 2)
        r2.close()
        w.close()
    finally:
        mock.ctl.mock_psrdada_destroy(0x51)


def test_process_baseband_on_ring_keys(mock, tmp_path, monkeypatch):
    """`process_baseband -k 40 -K 42 -C 46` (scripts/start_process:50): all three endpoints are psrdada rings"""
    # (dada_db -k 40 -b <one second> -n 16; the output rings hold the whole run: they are read afterwards)
    for key, bufsz, nbufs in ((0x40, 2 * FPS * 5032, 16), (0x42, 10 * SEG * TRIM, 8), (0x46, TRIM, 256)):
        assert mock.ctl.mock_psrdada_create(key, bufsz, nbufs) == 0
    monkeypatch.setattr(dada, "_SHIM", mock.shim)
    try:
        fr = _frames(13)
        feeder = dada.open_ring(0x40, "w")                       # stands in for writer (src/writer.c)
        feeder.write_header(_header())
        feeder.write(np.frombuffer(b"".join(f.tobytes() for f in fr), np.uint8))
        feeder.end_of_data()
        mock.ctl.mock_psrdada_shutdown(0x40)
        args = pbmod.build_parser().parse_args(
            ["-k", "40", "-K", "42", "-C", "46", "-b", "8", "-w", "2", "-r", "2", "-g", "0", "-o", "--datadir", str(tmp_path),
             "--logdir", str(tmp_path / "logs"), "--no-control", "--rows-per-seg", "8"])
        assert args.key_in == 0x40 and args.key_out == 0x42 and args.key_co == 0x46
        assert pbmod.run(args, handle=FakeHandle(nsets=2)) == 0
        raw, kur = _expected(fr, 12)
        co = dada.open_ring(0x46, "r")
        ch = vdif.ascii_header_parse(co.next_header())
        assert ch["NCHAN"] == "4096" and ch["SIGPROC_FILE"].endswith("_muos_ea99_kur.fil")
        assert co.read(len(kur) + 1) == kur                      # every segment of the excised stream
        out = dada.open_ring(0x42, "r")
        oh = vdif.ascii_header_parse(out.next_header())
        assert oh["SIGPROC_FILE"].endswith("_muos_ea07_kur.fil") and oh["NBIT"] == "8"
        assert out.read(len(kur) + 1) == kur                     # 10 s in one write, then second 11 and 12
        # the coadd ring got one write per segment (one buffer each) + end of data; the heimdall ring's buffers hold 10 s:
        # the first 10-s write fills one, seconds 11 and 12 share the end-of-data buffer
        assert mock.counts(0x46) == (12 * SEG + 1, 12 * SEG + 1, 1, 1) and mock.counts(0x42) == (2, 2, 1, 1)
    finally:
        for key in (0x40, 0x42, 0x46):
            mock.ctl.mock_psrdada_destroy(key)


def test_file_ring_parallel_readinto(tmp_path, monkeypatch):
    """Large reads of a regular file are split over threads (pread at explicit offsets): same bytes, same
    file position afterwards, short only at end of data; a FIFO keeps the sequential path."""
    import numpy as np
    rng = np.random.default_rng(5)
    body = rng.integers(0, 256, 5 * 1024 * 1024 + 123, dtype=np.uint8)
    path = tmp_path / "dump.vdif"
    path.write_bytes(b"H" * dada.DADA_HDR_SIZE + body.tobytes())
    monkeypatch.setattr(dada.FileRing, "_PAR_CHUNK", 256 * 1024)
    ring = dada.FileRing(str(path))
    assert ring.next_header() == b"H" * dada.DADA_HDR_SIZE
    first = ring.read(1000)
    assert first == body[:1000].tobytes()
    buf = np.zeros(3 * 1024 * 1024, np.uint8)
    assert ring.readinto(buf) == buf.size and ring._pool is not None
    assert np.array_equal(buf, body[1000:1000 + buf.size])
    rest = np.zeros(4 * 1024 * 1024, np.uint8)
    got = ring.readinto(rest)
    assert got == body.size - 1000 - buf.size
    assert np.array_equal(rest[:got], body[1000 + buf.size:])
    assert ring.readinto(rest) == 0


def test_shim_under_sanitizers(psrdada_mock, tmp_path):
    """pb_dada_shim.c compiled with -fsanitize=address,undefined (and once more with -fsanitize=thread) into
    tests/mock_psrdada/shim_driver.c: connect failure, both read interfaces, the empty end-of-data buffer, a stream
    four times the ring written by a second thread while 8 copy threads read it, close with a lock held."""
    mdir = psrdada_mock.mdir
    srcs = [os.path.join(mdir, "shim_driver.c"), os.path.join(ROOT, "vlite-fast_amd", "csrc", "pb_dada_shim.c"),
            os.path.join(mdir, "mock_psrdada.c")]
    for name, flags in (("asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]),
                        ("tsan", ["-fsanitize=thread"])):
        exe = str(tmp_path / ("shim_driver_" + name))
        subprocess.run(["gcc", "-g", "-O1", "-std=gnu11", "-Wall", "-Wextra", "-Werror"] + flags +
                       ["-I" + os.path.join(mdir, "include"), "-I" + os.path.join(ROOT, "include"), "-o", exe] + srcs + ["-lpthread"],
                       check=True)
        r = subprocess.run([exe], env=dict(os.environ, MOCK_PSRDADA_DIR=str(tmp_path)), stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=300)
        assert r.returncode == 0 and b"shim_driver: ok" in r.stdout, (name, r.stderr.decode()[-3000:])
        assert b"Sanitizer" not in r.stderr and b"runtime error" not in r.stderr, (name, r.stderr.decode()[-3000:])
