"""Candidates to the coincidencer: the TCP leg heimdall's `-coincidencer host:port` plays in the reference
(scripts/start_heimdall_single_antenna:21 `-coincidencer vlite-nrl:27555`; server side
/root/reference/src/trigger.py:7-8,37-43,84-123).

Wire format, as that server parses it: one TCP connection per gulp, the client writes text and closes
(the server reads until recv() returns 0, :95-100, then splits on newlines and drops empty lines):
    line 0   "<utc> <x> <y> <beam>"      utc = toks[0] ('%Y-%m-%d-%H:%M:%S', :170), beam = int(toks[3]), 1-based (:119)
    line 1   (one more header line, skipped: candidates start at lines[2], :128)
    line 2.. one candidate per line, the nine columns of src/candidate.py:8-18
             S/N, peak index, peak time, filter index, DM index, DM, members, first sample, last sample
A message of exactly two lines means "no candidates in this gulp" (:111-112).
heimdall itself (third party, absent) defines what x, y and line 1 hold; the coincidencer reads neither,
so they are written as the gulp's first sample and sample count and a column legend.
"""
import socket

HEIMDALL_PORT = 27555          # src/trigger.py:8


def candidate_line(c):
    """One heimdall-format text line (columns of src/candidate.py:8-18)."""
    return "%.6f\t%d\t%.6f\t%d\t%d\t%.4f\t%d\t%d\t%d" % (c["snr"], c["peak_idx"], c["peak_time"], c["tfilt"],
                                                        c["dmi"], c["dm"], c["ngiant"], c["i0"], c["i1"])


def format_message(utc_start, beam, cands, first_sample=0, nsamps=0):
    """The text of one gulp's message.  utc_start: the observation's UTC_START string (ring header key);
    beam: 1-based beam / antenna number as heimdall's -beam."""
    lines = ["%s %d %d %d" % (utc_start, first_sample, nsamps, beam),
             "# S/N peak_idx peak_time tfilt dmi dm ngiant i0 i1"]
    lines += [candidate_line(c) for c in cands]
    return ("\n".join(lines) + "\n").encode()


def send_candidates(host, port, utc_start, beam, cands, first_sample=0, nsamps=0, timeout=2.0):
    """Connect, write one gulp's message, close (the close is the end-of-message marker).
    Returns the number of bytes sent; raises OSError when the coincidencer is not reachable."""
    msg = format_message(utc_start, beam, cands, first_sample, nsamps)
    with socket.create_connection((host, port), timeout=timeout) as s:
        s.sendall(msg)
        s.shutdown(socket.SHUT_WR)
    return len(msg)


def parse_coincidencer(spec):
    """'host:port' as heimdall's -coincidencer takes it"""
    host, _, port = spec.rpartition(":")
    if not host:
        return spec, HEIMDALL_PORT
    return host, int(port)
