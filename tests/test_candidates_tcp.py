"""Candidate -> coincidencer TCP leg.  The server below is a restatement of the reference's
(src/trigger.py:37-43 make_server, :87-128 the accept / recv-until-close / split / parse loop, Python 2
there) so that the client is tested against what the coincidencer actually does with the bytes."""
import importlib
import socket
import threading

cand = importlib.import_module("vlite-fast_amd.candidates")


def _reference_server(sock, results):
    """src/trigger.py:90-128 for one connection"""
    clientsocket, _ = sock.accept()
    payload = []
    while True:
        msg = clientsocket.recv(4096)
        if len(msg) == 0:
            break
        payload.append(msg)
    clientsocket.close()
    lines = [l for l in (x.strip() for x in b"".join(payload).decode().split("\n")) if len(l) > 0]
    if len(lines) == 2:
        results.append(("empty", lines))
        return
    toks = lines[0].split()
    utc, beam = toks[0], int(toks[3]) - 1
    parsed = []
    for line in lines[2:]:                   # Candidate.__init__, src/candidate.py:5-18
        t = line.split()
        parsed.append(dict(sn=float(t[0]), peak_idx=int(t[1]), peak_time=float(t[2]), tfilt=int(t[3]), dmi=int(t[4]),
                           dm=float(t[5]), ngiant=int(t[6]), i0=int(t[7]), i1=int(t[8])))
    results.append((utc, beam, parsed))


def _serve():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    s.bind(("127.0.0.1", 0))
    s.listen(18)
    return s


def test_candidates_reach_the_coincidencer_as_it_parses_them():
    srv = _serve()
    results = []
    th = threading.Thread(target=_reference_server, args=(srv, results), daemon=True)
    th.start()
    cands = [dict(snr=12.345678, peak_idx=30123, peak_time=23.533594, tfilt=2, dmi=250, dm=500.1234, ngiant=17, i0=30120, i1=30127),
             dict(snr=7.5, peak_idx=1, peak_time=0.000781, tfilt=0, dmi=0, dm=2.0, ngiant=1, i0=1, i1=2)]
    n = cand.send_candidates("127.0.0.1", srv.getsockname()[1], "2016-07-01-01:00:00", 8, cands, first_sample=30720, nsamps=30720)
    th.join(timeout=5)
    assert not th.is_alive() and n > 0
    utc, beam, parsed = results[0]
    assert utc == "2016-07-01-01:00:00" and beam == 7                 # heimdall's beam is 1-based
    assert len(parsed) == 2
    assert parsed[0] == dict(sn=12.345678, peak_idx=30123, peak_time=23.533594, tfilt=2, dmi=250, dm=500.1234,
                             ngiant=17, i0=30120, i1=30127)
    assert parsed[1]["dm"] == 2.0 and parsed[1]["i1"] - parsed[1]["i0"] == 1
    srv.close()


def test_a_gulp_without_candidates_is_two_lines():
    srv = _serve()
    results = []
    th = threading.Thread(target=_reference_server, args=(srv, results), daemon=True)
    th.start()
    cand.send_candidates("127.0.0.1", srv.getsockname()[1], "2016-07-01-01:00:00", 1, [])
    th.join(timeout=5)
    assert results[0][0] == "empty" and len(results[0][1]) == 2
    srv.close()


def test_coincidencer_spec():
    assert cand.parse_coincidencer("vlite-nrl:27555") == ("vlite-nrl", 27555)
    assert cand.parse_coincidencer("vlite-nrl") == ("vlite-nrl", cand.HEIMDALL_PORT)


def test_overlap_rule_and_point_grouping():
    """search.overlap restates the reference coincidencer's test (src/candidate.py:49-65): fractional DM difference,
    width ratio <= 3, overlapping [i0, i1); search.group_points clusters the points of one series by time and DM
    (strongest first, deterministic under ties) and applies the width test only on request."""
    import numpy as np
    search = importlib.import_module("vlite-fast_amd.search")
    a = dict(dm=100.0, i0=1000, i1=1008)
    assert search.overlap(a, dict(dm=105.0, i0=1004, i1=1012))
    assert not search.overlap(a, dict(dm=112.0, i0=1004, i1=1012))               # |100/112 - 1| = 0.107 > 0.1
    assert search.overlap(dict(dm=112.0, i0=1004, i1=1012), dict(dm=102.0, i0=1000, i1=1008))   # relative to the OTHER's DM
    assert not search.overlap(a, dict(dm=100.0, i0=1008, i1=1016))               # [i0, i1) touch, do not overlap
    assert search.overlap(a, dict(dm=100.0, i0=1007, i1=1031))                   # widths 8 and 24: ratio 3 passes
    assert not search.overlap(a, dict(dm=100.0, i0=1007, i1=1032))               # 25 / 8 > 3
    assert search.overlap(a, dict(dm=100.0, i0=1007, i1=1032), delta_w=None)
    # one pulse seen at widths 1..64 around sample 500 at DM index 10, a second event far away in DM
    dms = np.arange(40) * 10.0
    idm = np.array([10, 10, 10, 10, 11, 30, 30])
    it = np.array([500, 499, 497, 480, 500, 505, 505])
    w = np.array([4, 8, 16, 64, 4, 2, 4])
    sn = np.array([20.0, 18.0, 15.0, 9.0, 12.0, 8.0, 8.0])
    c = search.group_points(idm, it, sn, w, dms, 7.8125e-4)
    assert [x["dmi"] for x in c] == [10, 30] and c[0]["ngiant"] == 5 and c[0]["i0"] == 480 and c[0]["i1"] == 544
    assert c[0]["tfilt"] == 2 and c[1]["ngiant"] == 2 and c[1]["tfilt"] == 1     # the tie at DM 30: the narrower first
    c3 = search.group_points(idm, it, sn, w, dms, 7.8125e-4, delta_w=3)
    assert len(c3) == 4 and c3[0]["ngiant"] == 3                                 # widths 16 and 64 split off
    pk = dict(dmi=idm[::-1].copy(), t=it[::-1].copy(), snr=sn[::-1].copy(), width_log2=np.log2(w[::-1]).astype(np.uint8))
    assert search.candidates_from_peaks(pk, dms, 7.8125e-4) == c                 # arrival order does not matter
