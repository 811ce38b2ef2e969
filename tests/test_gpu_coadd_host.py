"""BASELINE configs[3] end to end on the GPU: the coadder host (vlite-fast_amd/coadd_host.py) started as 2, 3 or 4
ranks (`--ranks N`, gloo rehearsal back end, all ranks on the one GPU this pool hands out) and as EIGHT ranks x SIXTEEN
antennas, two per rank -- configs[3]'s real shape -- with the ranks as threads of one process (`--dist-backend threads`,
vlite-fast_amd/threaded_ranks.py: the pool lets at most six processes use a card, so eight gloo processes are not an
option there), the rank's antennas batched in one handle -- pb_submit_vdif -> pb_process -> pb_coadd_local_tree ->
dist.gather -> pb_coadd_tree + pb_coadd_finish on the root -- against the oracle:
  * every antenna's own .fil / _kur.fil byte for byte = header + the oracle's codes of that antenna's data;
  * the ONE coadded file `..._ea99_kur.fil` byte for byte = the station-99 SIGPROC header +
    sel_and_dig( S(0, 1) * float(1/sqrt N) ) of the oracle's fp32 excised planes, S the DEFINED order of the fp32
    additions (antennas split by index parity, recursively: helpers.parity_sum) -- the same
    expectation whatever the number of ranks;
  * the coadded ring stand-in carries the same bytes behind a header naming that file.
(With RCCL the same code runs `dist.gather` on the device buffers; more than one RCCL rank needs more than one GPU.)
Reference: scripts/start_coadd:16,20-58; src/process_baseband.cu:272-285,1416-1422."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import count_tree, make_input, oracle_run, parity_sum

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vdif = importlib.import_module("vlite-fast_amd.vdif")
sigproc = importlib.import_module("vlite-fast_amd.sigproc")

R, SEG = 8, 10
NANT = 4
STATIONS = [3, 8, 11, 27, 30, 41, 52, 63, 5, 9, 14, 22, 33, 47, 58, 61]


def _dump(path, data, station):
    nsec = data.shape[0] // SEG
    hdr = vdif.writer_header(station, 0.8718, -0.72452, "B0833-45", 58000.0, "19A-331", 33, 3600)
    with open(path, "wb") as f:
        f.write(vdif.ascii_header_format(hdr))
        for s in range(nsec):
            p0 = np.concatenate([data[s * SEG + i, 0] for i in range(SEG)])
            p1 = np.concatenate([data[s * SEG + i, 1] for i in range(SEG)])
            f.write(vdif.frame_block(p0, p1, 3600 + s, 33, station).tobytes())


@pytest.mark.parametrize("nbit,NANT,ranks,order,layout,transport", [
    (8, 16, 8, "tree", "auto", "threads"), (8, 16, 8, "tree", "root", "threads"),
    (8, 8, 4, "tree", "auto", "gloo"), (8, 8, 4, "tree", "root", "gloo"), (8, 4, 2, "tree", "auto", "gloo"),
    (2, 3, 2, "tree", "auto", "gloo"), (4, 4, 2, "tree", "sliced", "gloo"), (8, 5, 3, "tree", "auto", "gloo"),
    (8, 4, 2, "fast", "auto", "gloo"), (2, 6, 4, "tree", "auto", "threads")])
def test_ranks_antennas_coadded_fil_is_byte_exact(tmp_path, oracle, nbit, NANT, ranks, order, layout, transport):
    """(8 ranks x 16 antennas, two per rank: configs[3] at its real shape, both layouts, the ranks as threads of one
    process; 4 ranks x 8 antennas: the same at half size with gloo processes; 2 ranks x 4: scale exactly 1/2; 2 bit,
    3 antennas: ranks hold {0, 2} and {1}, scale float(1 / sqrt 3); 3 ranks x 5 antennas: not a power of two, every
    antenna's plane goes to the root; "fast": one dist.reduce of left-to-right local sums, whose two-rank association
    is known.  layout: with a power-of-two world and one output polarisation the root's work is spread over the ranks
    ("auto" -> "sliced": all-to-all of plane slices, every rank sums and requantises its share, code bytes gathered);
    "root" gathers every plane to rank 0.  The same bytes either way, at 8, 4 and 2 bits.)"""
    nsec = 3 if NANT > 4 else 4                      # -> 2 / 3 s out (the last second of every stream is dropped)
    data = [make_input(80 + a, R, nsec * SEG, rfi=a != 1, dropped=a == 2) for a in range(NANT)]
    dumps = []
    for a in range(NANT):
        p = str(tmp_path / ("ant%d.uw" % a))
        _dump(p, data[a], STATIONS[a])
        dumps.append(p)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    cmd = [sys.executable, os.path.join(ROOT, "vlite-fast_amd", "coadd_host.py"), "--ranks", str(ranks), "--dist-backend", transport,
           "--share-gpus", "--coadd-order", order, "--coadd-layout", layout, "--replay"] + dumps + ["-b", str(nbit), "-r", "2", "-w", "2", "--datadir", str(tmp_path),
           "--logdir", str(tmp_path / "logs"), "--rows-per-seg", str(R), "--out-sink", str(tmp_path / "co_ring.bin")]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    nseg = (nsec - 1) * SEG
    dmjd = 57570 + 3600 / 86400.
    planes = []
    for a in range(NANT):
        res, _, _ = oracle_run(oracle, data[a][:nseg], R, rfi_mode=2, npol=1, nbit=nbit)
        hdr = sigproc.sigproc_header(STATIONS[a], 0.8718, -0.72452, "B0833-45", dmjd, 1, nbit)
        fb = tmp_path / ("20160701_010000_muos_ea%02d.fil" % STATIONS[a])
        fbk = tmp_path / ("20160701_010000_muos_ea%02d_kur.fil" % STATIONS[a])
        assert fb.read_bytes() == hdr + b"".join(x.codes_raw.tobytes() for x in res), "antenna %d raw" % a
        assert fbk.read_bytes() == hdr + b"".join(x.codes_kur.tobytes() for x in res), "antenna %d excised" % a
        planes.append([x.ave_kur for x in res])
    scale = np.float32(1.0 / np.sqrt(float(NANT)))
    want = b""
    naive = b""
    for s in range(nseg):
        if order == "tree":
            tot = parity_sum([planes[a][s] for a in range(NANT)])
        else:
            tot = ((np.float32(0) + planes[0][s]) + planes[2][s]) + ((np.float32(0) + planes[1][s]) + planes[3][s])
        want += oracle.sel_and_dig(tot * scale, R, npol=1, nbit=nbit).tobytes()
        lr = np.float32(0)
        for a in range(NANT):
            lr = lr + planes[a][s]
        naive += oracle.sel_and_dig(lr * scale, R, npol=1, nbit=nbit).tobytes()
    co_hdr = sigproc.sigproc_header(99, 0.8718, -0.72452, "B0833-45", dmjd, 1, nbit)
    co = (tmp_path / "20160701_010000_muos_ea99_kur.fil").read_bytes()
    assert co[:len(co_hdr)] == co_hdr
    got = np.frombuffer(co[len(co_hdr):], np.uint8)
    assert got.size == len(want)
    assert co[len(co_hdr):] == want, "%d coadded bytes differ" % int((got != np.frombuffer(want, np.uint8)).sum())
    ring = (tmp_path / "co_ring.bin").read_bytes()
    rh = vdif.ascii_header_parse(ring[:4096])
    assert rh["STATIONID"] == "99" and rh["SIGPROC_FILE"].endswith("_muos_ea99_kur.fil") and rh["NBIT"] == str(nbit)
    assert ring[4096:] == want
    # the coadded stream is not any single antenna's
    assert want != b"".join(oracle.sel_and_dig(planes[0][s], R, nbit=nbit).tobytes() for s in range(nseg))
    log = "".join(open(os.path.join(str(tmp_path / "logs"), f)).read() for f in os.listdir(str(tmp_path / "logs")))
    assert 'order "%s"' % order in log
    want_layout = "sliced" if (order == "tree" and ranks in (2, 4, 8) and layout != "root") else "root"
    assert ("rank %d of %d" % (ranks - 1, ranks)) in log
    assert 'layout "%s"' % want_layout in log


def test_single_rank_coadd_of_two_antennas_equals_two_rank_sum(tmp_path, oracle):
    """One rank holding both antennas (local sum only, no collective): the same host, world size 1."""
    nsec = 3
    data = [make_input(90 + a, R, nsec * SEG) for a in range(2)]
    dumps = []
    for a in range(2):
        p = str(tmp_path / ("ant%d.uw" % a))
        _dump(p, data[a], STATIONS[a])
        dumps.append(p)
    cmd = [sys.executable, os.path.join(ROOT, "vlite-fast_amd", "coadd_host.py"), "--replay"] + dumps + [
        "-b", "8", "-r", "2", "-w", "0", "--datadir", str(tmp_path), "--logdir", str(tmp_path / "logs"), "--rows-per-seg", str(R)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    nseg = (nsec - 1) * SEG
    planes = [[x.ave_kur for x in oracle_run(oracle, data[a][:nseg], R)[0]] for a in range(2)]
    scale = np.float32(1.0 / np.sqrt(2.0))
    want = b"".join(oracle.sel_and_dig(parity_sum([planes[0][s], planes[1][s]]) * scale, R).tobytes() for s in range(nseg))
    co_hdr = sigproc.sigproc_header(99, 0.8718, -0.72452, "B0833-45", 57570 + 3600 / 86400., 1, 8)
    assert (tmp_path / "20160701_010000_muos_ea99_kur.fil").read_bytes() == co_hdr + want
    assert not (tmp_path / ("20160701_010000_muos_ea%02d.fil" % STATIONS[0])).exists()      # -w 0


@pytest.mark.parametrize("rfi_mode,npol", [(0, 1), (2, 2)])
def test_single_rank_coadd_other_modes(tmp_path, oracle, rfi_mode, npol):
    """RFI mode 0 (the raw stream is what is summed; the coadded file is `_ea99.fil`) and -P 2 (two-pol output:
    [time][pol][channel] bytes, nifs = 2): two antennas on one rank against the oracle."""
    nsec = 3
    data = [make_input(95 + a, R, nsec * SEG) for a in range(2)]
    dumps = []
    for a in range(2):
        p = str(tmp_path / ("ant%d.uw" % a))
        _dump(p, data[a], STATIONS[a])
        dumps.append(p)
    cmd = [sys.executable, os.path.join(ROOT, "vlite-fast_amd", "coadd_host.py"), "--replay"] + dumps + [
        "-b", "8", "-r", str(rfi_mode), "-P", str(npol), "-w", "2", "--datadir", str(tmp_path), "--logdir", str(tmp_path / "logs"),
        "--rows-per-seg", str(R)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    nseg = (nsec - 1) * SEG
    res = [oracle_run(oracle, data[a][:nseg], R, rfi_mode=rfi_mode, npol=npol, nbit=8)[0] for a in range(2)]
    key = "ave_raw" if rfi_mode == 0 else "ave_kur"
    scale = np.float32(1.0 / np.sqrt(2.0))
    want = b"".join(oracle.sel_and_dig(parity_sum([getattr(res[0][s], key), getattr(res[1][s], key)]) * scale, R, npol=npol).tobytes()
                    for s in range(nseg))
    co_hdr = sigproc.sigproc_header(99, 0.8718, -0.72452, "B0833-45", 57570 + 3600 / 86400., npol, 8)
    name = "20160701_010000_muos_ea99%s.fil" % ("" if rfi_mode == 0 else "_kur")
    assert (tmp_path / name).read_bytes() == co_hdr + want
    # the antennas' own files as process_baseband writes them in that mode
    hdr = sigproc.sigproc_header(STATIONS[0], 0.8718, -0.72452, "B0833-45", 57570 + 3600 / 86400., npol, 8)
    own = tmp_path / ("20160701_010000_muos_ea%02d%s.fil" % (STATIONS[0], ""))
    assert own.read_bytes() == hdr + b"".join(x.codes_raw.tobytes() for x in res[0])


def test_fullsize_coadd_two_antennas_one_second(tmp_path, oracle):
    """R = 1024 (the production segment; the two-antennas-per-GPU shape of configs[3]): two 2-s dumps -> one coadded
    second, against the oracle's 2 x 10 full-size segments."""
    Rf, nsec = 1024, 2
    data = [make_input(60 + a, Rf, nsec * SEG, rfi=a == 0, dropped=a == 1) for a in range(2)]
    dumps = []
    for a in range(2):
        p = str(tmp_path / ("ant%d.uw" % a))
        hdr = vdif.writer_header(STATIONS[a], 0.8718, -0.72452, "B0833-45", 58000.0, "19A-331", 33, 3600)
        with open(p, "wb") as f:
            f.write(vdif.ascii_header_format(hdr))
            for s in range(nsec):
                p0 = np.concatenate([data[a][s * SEG + i, 0] for i in range(SEG)])
                p1 = np.concatenate([data[a][s * SEG + i, 1] for i in range(SEG)])
                f.write(vdif.frame_block(p0, p1, 3600 + s, 33, STATIONS[a]).tobytes())
        dumps.append(p)
    cmd = [sys.executable, os.path.join(ROOT, "vlite-fast_amd", "coadd_host.py"), "--replay"] + dumps + [
        "-b", "8", "-r", "2", "-w", "2", "--datadir", str(tmp_path), "--logdir", str(tmp_path / "logs")]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    nseg = (nsec - 1) * SEG
    res = [oracle_run(oracle, data[a][:nseg], Rf)[0] for a in range(2)]
    scale = np.float32(1.0 / np.sqrt(2.0))
    want = b"".join(oracle.sel_and_dig(parity_sum([res[0][s].ave_kur, res[1][s].ave_kur]) * scale, Rf).tobytes() for s in range(nseg))
    dmjd = 57570 + 3600 / 86400.
    co_hdr = sigproc.sigproc_header(99, 0.8718, -0.72452, "B0833-45", dmjd, 1, 8)
    assert (tmp_path / "20160701_010000_muos_ea99_kur.fil").read_bytes() == co_hdr + want
    for a in range(2):
        hdr = sigproc.sigproc_header(STATIONS[a], 0.8718, -0.72452, "B0833-45", dmjd, 1, 8)
        assert (tmp_path / ("20160701_010000_muos_ea%02d_kur.fil" % STATIONS[a])).read_bytes() == hdr + b"".join(x.codes_kur.tobytes() for x in res[a])
        assert (tmp_path / ("20160701_010000_muos_ea%02d.fil" % STATIONS[a])).read_bytes() == hdr + b"".join(x.codes_raw.tobytes() for x in res[a])


def _levels(codes, nbit, npol, ntime):
    """bytes in sel_and_dig order -> fp32 cell centres in the planes' layout (include/pb_hip.h: pb_coadd_local_codes)"""
    b = np.frombuffer(codes, np.uint8) if not isinstance(codes, np.ndarray) else codes.reshape(-1)
    per = 8 // nbit
    q = np.stack([(b >> (nbit * j)) & ((1 << nbit) - 1) for j in range(per)], axis=1).reshape(-1).astype(np.float64)
    if nbit == 8:
        v = ((q - 127.0) * 0.02957).astype(np.float32)
    elif nbit == 4:
        v = ((q - 7.0) * 0.3188).astype(np.float32)
    else:
        lv = np.array([-0.6109 - 0.503975, 0.5 * (-0.6109 + 0.3970), 0.5 * (0.3970 + 1.4050), 1.4050 + 0.503975]).astype(np.float32)
        v = lv[q.astype(np.int64)]
    if npol == 2:
        v = v.reshape(ntime, 2, 4096).transpose(1, 0, 2).reshape(-1)
    return v


@pytest.mark.parametrize("nbit,npol,ranks,nant", [(8, 1, 2, 3), (2, 1, 1, 2), (4, 2, 1, 2)])
def test_coadd_from_quantised_codes(tmp_path, oracle, nbit, npol, ranks, nant):
    """`--coadd-input codes` (the like-for-like mode): what is summed is each antenna's quantised
    filterbank -- the bytes of its own _kur.fil, each code standing for the centre of its quantiser cell -- not the
    fp32 planes.  Expected from the oracle's codes alone; ranks hold {0, 2} and {1}."""
    nsec = 3
    data = [make_input(70 + a, R, nsec * SEG, rfi=a != 1) for a in range(nant)]
    dumps = []
    for a in range(nant):
        p = str(tmp_path / ("ant%d.uw" % a))
        _dump(p, data[a], STATIONS[a])
        dumps.append(p)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    cmd = [sys.executable, os.path.join(ROOT, "vlite-fast_amd", "coadd_host.py")]
    if ranks > 1:
        cmd += ["--ranks", str(ranks), "--dist-backend", "gloo", "--share-gpus"]
    cmd += ["--coadd-input", "codes", "--replay"] + dumps + ["-b", str(nbit), "-P", str(npol), "-r", "2", "-w", "2",
            "--datadir", str(tmp_path), "--logdir", str(tmp_path / "logs"), "--rows-per-seg", str(R)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    nseg = (nsec - 1) * SEG
    res = [oracle_run(oracle, data[a][:nseg], R, rfi_mode=2, npol=npol, nbit=nbit)[0] for a in range(nant)]
    dmjd = 57570 + 3600 / 86400.
    for a in range(nant):                  # the antennas' own files are what they always are
        hdr = sigproc.sigproc_header(STATIONS[a], 0.8718, -0.72452, "B0833-45", dmjd, npol, nbit)
        fbk = tmp_path / ("20160701_010000_muos_ea%02d_kur.fil" % STATIONS[a])
        assert fbk.read_bytes() == hdr + b"".join(x.codes_kur.tobytes() for x in res[a])
    ntime = R // 8
    scale = np.float32(1.0 / np.sqrt(float(nant)))
    want, fp32 = b"", b""
    for s in range(nseg):
        lv = [_levels(res[a][s].codes_kur, nbit, npol, ntime) for a in range(nant)]
        if ranks == 2:
            tot = ((np.float32(0) + lv[0]) + lv[2]) + (np.float32(0) + lv[1])
        else:
            tot = (np.float32(0) + lv[0]) + lv[1]
        full = np.zeros((npol, ntime, oracle.NCHAN), np.float32)      # the oracle's sel_and_dig takes the full band
        full[:, :, oracle.CHANMIN:oracle.CHANMIN + 4096] = (tot * scale).reshape(npol, ntime, 4096)
        want += oracle.sel_and_dig(full, R, npol=npol, nbit=nbit).tobytes()
        p = np.float32(0) + res[0][s].ave_kur
        for a in range(1, nant):
            p = p + res[a][s].ave_kur
        fp32 += oracle.sel_and_dig(p * scale, R, npol=npol, nbit=nbit).tobytes()
    co_hdr = sigproc.sigproc_header(99, 0.8718, -0.72452, "B0833-45", dmjd, npol, nbit)
    co = (tmp_path / "20160701_010000_muos_ea99_kur.fil").read_bytes()
    assert co[:len(co_hdr)] == co_hdr and len(co) - len(co_hdr) == len(want)
    got = np.frombuffer(co[len(co_hdr):], np.uint8)
    assert co[len(co_hdr):] == want, "%d coadded bytes differ" % int((got != np.frombuffer(want, np.uint8)).sum())
    # and it is a different product from the fp32 sum: quantising twice loses what the planes still had
    assert want != fp32
    if nbit == 8:
        d = np.abs(got.astype(np.int32) - np.frombuffer(fp32, np.uint8).astype(np.int32))
        # (within 2 codes wherever no antenna's byte had clipped; a clipped 0 / 255 has lost its excess for good)
        assert (d <= 2).mean() > 0.95 and 0.0 < (d > 0).mean() < 0.8
