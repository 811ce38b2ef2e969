"""The native host program (vlite-fast_amd/csrc/process_baseband, C++ above the C ABI): header bytes and file
names against the Python host formats without a GPU; on the GPU the same replay checks as the Python
executable (byte-identical .fil files against the oracle, sinks, dropped frames, the 10-s output ring)."""
import importlib
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "vlite-fast_amd", "csrc", "process_baseband")
vdif = importlib.import_module("vlite-fast_amd.vdif")
sigproc = importlib.import_module("vlite-fast_amd.sigproc")

R = 8
SEG = 10


def _ensure_built():
    """the program is built by __graft_entry__.build(); a tree that only carries the library builds it here (g++)"""
    if not os.access(EXE, os.X_OK):
        subprocess.run(["make", "-s", "-C", os.path.dirname(EXE), "process_baseband"], check=False)


def _run(argv, **kw):
    _ensure_built()
    return subprocess.run([EXE] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, **kw)


def test_executable_is_built():
    _ensure_built()
    assert os.access(EXE, os.X_OK), "run __graft_entry__.build() (make -C vlite-fast_amd/csrc)"
    r = _run(["-h"])
    assert r.returncode == 0 and b"usage: process_baseband" in r.stdout
    assert _run(["-b", "3", "--replay", "/dev/null"]).returncode == 1          # Unsupported NBIT!


@pytest.mark.parametrize("station,ra,dec,name", [(7, 0.8718, -0.72452, "B0833-45"), (12, 5.2, 0.3, "J0534+22")])
def test_headers_and_names_equal_the_python_formats(tmp_path, station, ra, dec, name):
    inhdr = vdif.writer_header(station, ra, dec, name, 58000.25, "19A-331", 33, 3600)
    frame = np.zeros(vdif.VD_FRM, np.uint8)
    frame[:32] = vdif.pack_header(3600, 33, 0, station, 0).view(np.uint8)
    dump = tmp_path / "one.uw"
    dump.write_bytes(vdif.ascii_header_format(inhdr) + frame.tobytes())
    r = _run(["-b", "8", "-P", "1", "-r", "2", "--replay", str(dump), "--datadir", "/data/x", "--logdir", str(tmp_path),
              "--no-control", "--rows-per-seg", str(R), "--dump-headers"])
    assert r.returncode == 0, r.stderr
    out = dict((l.split(" ", 1)[0], l.split(" ", 1)[1]) for l in r.stdout.decode().splitlines())
    vh = vdif.unpack_header(frame.tobytes())
    fps = R * SEG * 12500 // 5000
    t_unix = vdif.vdif_to_unixepoch(vh)
    fb, fbk, co, cok = sigproc.fb_names(t_unix, station, "/data/x")
    assert out["files"].split() == [fb, fbk, co, cok]
    want_sp = sigproc.sigproc_header(station, ra, dec, name, vdif.frame_dmjd(vh, fps), 1, 8)
    n, hx = out["sigproc"].split()
    assert int(n) == len(want_sp) and bytes.fromhex(hx) == want_sp
    pin = vdif.ascii_header_parse(vdif.ascii_header_format(inhdr))
    for key, f in (("out", fbk), ("co", cok)):
        want = vdif.ascii_header_format(sigproc.psrdada_out_header(pin, vh, 1, 8, f, t_unix, vdif.frame_mjd(vh), vdif.frame_mjd_sec(vh)))
        n, hx = out[key].split()
        assert int(n) == 4096 and bytes.fromhex(hx) == want


def _dump(path, data, station=7, drop=()):
    nsec = data.shape[0] // SEG
    hdr = vdif.writer_header(station, 0.8718, -0.72452, "B0833-45", 58000.0, "19A-331", 33, 3600)
    with open(path, "wb") as f:
        f.write(vdif.ascii_header_format(hdr))
        for s in range(nsec):
            p0 = np.concatenate([data[s * SEG + i, 0] for i in range(SEG)])
            p1 = np.concatenate([data[s * SEG + i, 1] for i in range(SEG)])
            blk = vdif.frame_block(p0, p1, 3600 + s, 33, station).reshape(-1, vdif.VD_FRM)
            keep = [i for i in range(blk.shape[0]) if (s, i) not in drop]
            f.write(blk[keep].tobytes())


def _argv(tmp_path, dump, nbit, extra=()):
    return ["-k", "40", "-K", "0", "-w", "2", "-b", str(nbit), "-P", "1", "-r", "2", "-g", "0", "-p", "0",
            "--replay", dump, "--datadir", str(tmp_path), "--logdir", str(tmp_path / "logs"), "--no-control",
            "--rows-per-seg", str(R), "--co-sink", str(tmp_path / "co.bin"), "--out-sink", str(tmp_path / "out.bin")] + list(extra)


@pytest.mark.gpu
@pytest.mark.parametrize("nbit", [8, 2])
def test_native_replay_to_fil_is_byte_exact(tmp_path, oracle, nbit):
    from helpers import make_input, oracle_run
    nsec = 4
    data = make_input(11, R, nsec * SEG)
    dump = str(tmp_path / "obs.uw")
    _dump(dump, data)
    r = _run(_argv(tmp_path, dump, nbit, ["-t"]))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    fb = tmp_path / "20160701_010000_muos_ea07.fil"
    fbk = tmp_path / "20160701_010000_muos_ea07_kur.fil"
    nseg = (nsec - 1) * SEG                    # the reference drops the last second
    res, _, _ = oracle_run(oracle, data[:nseg], R, rfi_mode=2, npol=1, nbit=nbit)
    hdr = sigproc.sigproc_header(7, 0.8718, -0.72452, "B0833-45", 57570 + 3600 / 86400., 1, nbit)
    assert fb.read_bytes() == hdr + b"".join(r_.codes_raw.tobytes() for r_ in res)
    assert fbk.read_bytes() == hdr + b"".join(r_.codes_kur.tobytes() for r_ in res)
    co = (tmp_path / "co.bin").read_bytes()
    ch = vdif.ascii_header_parse(co[:4096])
    assert ch["NCHAN"] == "4096" and ch["NBIT"] == str(nbit) and ch["SIGPROC_FILE"].endswith("_muos_ea99_kur.fil")
    assert co[4096:] == b"".join(r_.codes_kur.tobytes() for r_ in res)
    assert len((tmp_path / "out.bin").read_bytes()) == 4096      # nothing before 10 s have been integrated
    log = next((tmp_path / "logs").glob("*_process_*.log")).read_text()
    for line in ("Wrote", "Proc Time...", "Read Time...", "Copy To Dev.", "Kurtosis....", "FFT.........", "Normalize...", "Write......."):
        assert line in log


@pytest.mark.gpu
def test_native_and_python_hosts_write_the_same_bytes(tmp_path):
    """12 s with dropped frames (one at a second's very start) through both hosts: .fil files, coadd sink and the
    output ring's 10-s-then-1-s writes are identical."""
    from helpers import make_input
    pbmod = importlib.import_module("vlite-fast_amd.process_baseband")
    nsec = 13
    data = make_input(17, R, nsec * SEG)
    dump = str(tmp_path / "obs.uw")
    _dump(dump, data, drop={(1, 57), (2, 0), (2, 1), (5, 399)})
    outs = {}
    for who in ("py", "cxx"):
        d = tmp_path / who
        d.mkdir()
        argv = _argv(d, dump, 8)
        if who == "py":
            assert pbmod.run(pbmod.build_parser().parse_args(argv)) == 0
        else:
            r = _run(argv)
            assert r.returncode == 0, r.stderr.decode()[-2000:]
        outs[who] = {n: (d / n).read_bytes() for n in ("20160701_010000_muos_ea07.fil", "20160701_010000_muos_ea07_kur.fil",
                                                       "co.bin", "out.bin")}
    for n in outs["py"]:
        a, b = outs["py"][n], outs["cxx"][n]
        if n.endswith(".bin"):       # ring stand-ins: the header names the .fil file, whose directory differs here
            ha, hb = vdif.ascii_header_parse(a[:4096]), vdif.ascii_header_parse(b[:4096])
            ha["SIGPROC_FILE"], hb["SIGPROC_FILE"] = os.path.basename(ha["SIGPROC_FILE"]), os.path.basename(hb["SIGPROC_FILE"])
            assert ha == hb and list(ha) == list(hb), n
            a, b = a[4096:], b[4096:]
        assert a == b, n
    assert len(outs["cxx"]["out.bin"]) == 4096 + 12 * SEG * (R // 8) * 4096      # 10 s at once, then 1 s twice


@pytest.mark.gpu
def test_native_cmd_quit(tmp_path):
    """H7: 'Q' on the control socket ends the run with status 0 and the reference's log line (src/utils.c:174-220,
    src/process_baseband.cu:1081); the command is looked at once per second of data."""
    import socket
    import threading
    import time
    from helpers import make_input
    data = make_input(19, R, 4 * SEG)
    dump = str(tmp_path / "obs.uw")
    _dump(dump, data)
    fifo = str(tmp_path / "ring.fifo")
    os.mkfifo(fifo)
    port = 20000 + (os.getpid() % 5000) + 7
    argv = [a for a in _argv(tmp_path, fifo, 8) if a != "--no-control"] + ["--control-port", str(port)]
    _ensure_built()
    proc = subprocess.Popen([EXE] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    try:
        logs = tmp_path / "logs"
        for _ in range(600):        # the handle is created first; then the program waits for a header
            if logs.exists() and any("Waiting for DADA header." in p.read_text() for p in logs.glob("*.log")):
                break
            time.sleep(0.1)
        socket.socket(socket.AF_INET, socket.SOCK_DGRAM).sendto(b"xQ", ("127.0.0.1", port))

        def feed():
            with open(fifo, "wb") as w, open(dump, "rb") as r:
                try:
                    w.write(r.read())
                except BrokenPipeError:
                    pass
        t = threading.Thread(target=feed, daemon=True)
        t.start()
        out, err = proc.communicate(timeout=120)
    finally:
        if proc.poll() is None:
            proc.kill()
    assert proc.returncode == 0, err.decode()[-2000:]
    log = next((tmp_path / "logs").glob("*_process_*.log")).read_text()
    assert "Received CMD_QUIT, indicating data taking is ceasing.  Exiting." in log


def test_ring_keys_reach_the_shim(tmp_path, psrdada_mock):
    """`-k 40` without --replay goes to the psrdada shim (dlopen at run time): without the library the program says
    what to build; with the REAL shim (built over the psrdada stand-in of tests/mock_psrdada; no ring 0x40 exists) it
    binds all nine symbols and reports the failed connect.  No GPU is touched before the rings are connected."""
    env = dict(os.environ, PB_DADA_LIB=str(tmp_path / "nope.so"))
    r = _run(["-k", "40", "-b", "8", "--logdir", str(tmp_path), "-o", "--no-control"], env=env)
    assert r.returncode == 1 and b"psrdada rings need the shim library" in r.stdout
    env = dict(os.environ, PB_DADA_LIB=psrdada_mock.shim_path, MOCK_PSRDADA_DIR=str(tmp_path))
    r = _run(["-k", "40", "-b", "8", "--logdir", str(tmp_path), "-o", "--no-control"], env=env)
    assert r.returncode == 1 and b"could not connect to input ring 40" in r.stdout and b"lacks symbols" not in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("threads", ["8", "1"])
def test_native_host_on_rings_equals_replay(tmp_path, psrdada_mock, threads):
    """`process_baseband -k 40 -K 42 -C 46` (scripts/start_process:50) as its own PROCESS on three rings, through the
    real shim (block-level reads with 8 copy threads / the reference's ipcio_read with PB_DADA_THREADS=1): a writer
    (this process, tools/dump_to_ring.py's calls) has filled ring 40 with a 13-s dump; afterwards the .fil files equal
    those of a --replay of the same dump, ring 46 holds every segment of the excised stream and ring 42 the 10-s-then-
    1-s writes.  The rings are the stand-in's (shared files): the shim's logic and the host's ring code are what is
    exercised, not psrdada."""
    from helpers import make_input
    dada = importlib.import_module("vlite-fast_amd.dada")
    nsec = 13
    data = make_input(19, R, nsec * SEG)
    dump = str(tmp_path / "obs.uw")
    _dump(dump, data, drop={(1, 57), (2, 0)})
    (tmp_path / "replay").mkdir()
    r = _run(_argv(tmp_path / "replay", dump, 8))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    fps = R * SEG * 12500 // 5000
    trim = 2 * R * 4096 // 16
    ctl, keys = psrdada_mock.ctl, ((0x40, 2 * fps * 5032, 16), (0x42, 10 * SEG * trim, 8), (0x46, trim, 256))
    rdir = tmp_path / "rings"
    rdir.mkdir()
    old = os.environ.get("MOCK_PSRDADA_DIR")
    os.environ["MOCK_PSRDADA_DIR"] = str(rdir)
    try:
        for key, bufsz, nbufs in keys:
            assert ctl.mock_psrdada_create(key, bufsz, nbufs) == 0
        raw = open(dump, "rb").read()
        w = dada.PsrdadaRing(0x40, "w", lib=psrdada_mock.shim)
        w.write_header(raw[:4096])
        w.write(np.frombuffer(raw[4096:], np.uint8))
        w.end_of_data()
        w.close()
        ctl.mock_psrdada_shutdown(0x40)
        (tmp_path / "ring").mkdir()
        env = dict(os.environ, PB_DADA_LIB=psrdada_mock.shim_path, PB_DADA_THREADS=threads)
        r = _run(["-k", "40", "-K", "42", "-C", "46", "-w", "2", "-b", "8", "-P", "1", "-r", "2", "-g", "0", "--datadir",
                  str(tmp_path / "ring"), "--logdir", str(tmp_path / "ring" / "logs"), "--no-control", "--rows-per-seg", str(R)], env=env)
        assert r.returncode == 0, (r.stdout.decode()[-2000:], r.stderr.decode()[-2000:])
        names = sorted(p.name for p in (tmp_path / "replay").glob("*.fil"))
        assert len(names) == 2
        for n in names:
            assert (tmp_path / "ring" / n).read_bytes() == (tmp_path / "replay" / n).read_bytes(), n
        kur = (tmp_path / "replay" / names[1]).read_bytes()
        hl = len(sigproc.sigproc_header(7, 0.8718, -0.72452, "B0833-45", 57570 + 3600 / 86400., 1, 8))
        co = dada.PsrdadaRing(0x46, "r", lib=psrdada_mock.shim)
        ch = vdif.ascii_header_parse(co.next_header())
        assert ch["SIGPROC_FILE"].endswith("_muos_ea99_kur.fil")
        assert co.read(len(kur)) == kur[hl:]
        co.close()
        out = dada.PsrdadaRing(0x42, "r", lib=psrdada_mock.shim)
        assert vdif.ascii_header_parse(out.next_header())["SIGPROC_FILE"].endswith("_muos_ea07_kur.fil")
        assert out.read(len(kur)) == kur[hl:]
        out.close()
        nb = (len(raw) - 4096) // (2 * fps * 5032) + 1           # full one-second buffers + the end-of-data one
        assert psrdada_mock.counts(0x40)[:2] == (nb, nb)         # the host has handed every buffer back
    finally:
        for key, _, _ in keys:
            ctl.mock_psrdada_destroy(key)
        if old is None:
            os.environ.pop("MOCK_PSRDADA_DIR", None)
        else:
            os.environ["MOCK_PSRDADA_DIR"] = old


@pytest.mark.gpu
@pytest.mark.parametrize("rfi_mode,nbit,npol", [(0, 8, 1), (1, 4, 1), (2, 2, 2)])
def test_native_and_python_hosts_agree_in_other_modes(tmp_path, rfi_mode, nbit, npol):
    from helpers import make_input
    pbmod = importlib.import_module("vlite-fast_amd.process_baseband")
    data = make_input(23, R, 4 * SEG)
    dump = str(tmp_path / "obs.uw")
    _dump(dump, data)
    got = {}
    for who in ("py", "cxx"):
        d = tmp_path / who
        d.mkdir()
        argv = ["-w", "2", "-b", str(nbit), "-P", str(npol), "-r", str(rfi_mode), "--replay", dump, "--datadir", str(d),
                "--logdir", str(d / "logs"), "--no-control", "--rows-per-seg", str(R), "--co-sink", str(d / "co.bin")]
        if who == "py":
            assert pbmod.run(pbmod.build_parser().parse_args(argv)) == 0
        else:
            r = _run(argv)
            assert r.returncode == 0, r.stderr.decode()[-2000:]
        got[who] = {p.name: p.read_bytes() for p in sorted(d.glob("*.fil"))}
        got[who]["co"] = (d / "co.bin").read_bytes()[4096:]
    assert list(got["py"]) == list(got["cxx"]) and len(got["py"]) == (3 if rfi_mode == 2 else 2)
    for n in got["py"]:
        assert got["py"][n] == got["cxx"][n] and len(got["py"][n]) > 1000, n
