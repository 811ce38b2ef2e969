#!/usr/bin/env python3
"""Host side of the reference's `process_baseband` executable, driving libpb_hip.so.

Drop-in under scripts/start_process:50 of the reference:
    process_baseband -k <bb_key> -K <fb_key> -w <0|1|2> -b <2|4|8> -g <gpu> -o -C <coadd_key>
(+ optional -P 1|2  -r 0|1|2  -i  -s  -t; the legacy `-p N` of scripts/baseband_test:26 is
accepted and ignored, as the reference's getopt string silently does).

Mirrors /root/reference/src/process_baseband.cu main():
  :358-470   option parsing (keys are hexadecimal: `-k 40` means 0x40)
  :784-1000  per-observation set-up: ring header, first frame, file names, SIGPROC headers
  :1015-1496 per-second loop: a second is dispatched only once a frame of the NEXT second has
             been seen, so the last second of every observation is dropped (:1058-1064)
  :1108-1458 ten 100-ms segments per second -> here ONE pb_submit_vdif + pb_process call
  :1416-1441 coadd ring (one write per segment), .fil / _kur.fil
  :1482-1494 output ring: the first write is the 10-s buffer, then one second per second
The bandpass state lives in the PbHandle and persists across observations (:700-709).

Deliberate differences: in RFI modes 0 and 1 the reference fwrite()s an
uninitialised host pointer (:678-682 vs :1438); here the one stream that exists is written.
"""
import argparse
import importlib
import os
import socket
import struct
import sys
import time

import numpy as np

_pkg = __package__ or "vlite-fast_amd"
if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
vdif = importlib.import_module(_pkg + ".vdif")
sigproc = importlib.import_module(_pkg + ".sigproc")
dada = importlib.import_module(_pkg + ".dada")

SEG_PER_SEC = 10
MC_GROUP, MC_READER_PORT = "224.3.29.71", 20000       # src/multicast.h:14,16
CMD_QUIT = ord("Q")                                   # src/def.h:6
LOGDIR = "/home/vlite-master/mtk/logs"                # src/def.h:26

# `-w 1` writes only for the sources a site file lists (site policy, src/util.c:91-152 of the reference; out of the
# hot path's scope, so it is data, not code: vlite-fast_amd/site/write_allow.txt or $PB_WRITE_ALLOW)
WRITE_ALLOW_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "site", "write_allow.txt")


def load_write_allow(path=None, log=None):
    """-> (NAME substrings, DATAID substrings, (ra, dec) positions [rad]); all empty when there is no file -- said
    in the log, with the path that was tried, so that `-w 1` never drops data silently"""
    path = path or os.environ.get("PB_WRITE_ALLOW") or WRITE_ALLOW_FILE
    names, ids, coords = [], [], []
    try:
        with open(path) as f:
            for line in f:
                t = line.split("#", 1)[0].split()
                if len(t) == 2 and t[0] == "name":
                    names.append(t[1])
                elif len(t) == 2 and t[0] == "dataid":
                    ids.append(t[1])
                elif len(t) == 3 and t[0] == "coords":
                    coords.append((float(t[1]), float(t[2])))
        if log:
            log("INFO", "Source list for -w 1: %s (%d names, %d dataids, %d positions)." % (path, len(names), len(ids), len(coords)))
    except (OSError, ValueError) as e:
        if log:
            log("ERR", "Source list for -w 1 not usable (%s: %s): no source will be recorded." % (path, e))
    return tuple(names), tuple(ids), tuple(coords)


COORD_TOL = 0.01      # check_coords' default tolerance [rad], src/util.h


def coord_dist(ra1, ra2, de1, de2):
    """src/util.c:127-134 (radians)"""
    import math
    dde = de2 - de1
    dra = (ra2 - ra1) * math.cos(de1)
    return math.sqrt(dde * dde + dra * dra)


def build_parser():
    p = argparse.ArgumentParser(prog="process_baseband", add_help=False)
    p.add_argument("-h", action="help")
    p.add_argument("-k", dest="key_in", type=lambda s: int(s, 16), default=0x40)
    p.add_argument("-K", dest="key_out", type=lambda s: int(s, 16), default=0)
    p.add_argument("-C", dest="key_co", type=lambda s: int(s, 16), default=0)
    p.add_argument("-o", dest="stdout_output", action="store_true")
    p.add_argument("-m", dest="muos", action="store_true")   # accepted, no effect (:403-407)
    p.add_argument("-i", dest="inject_frb", action="store_true")
    p.add_argument("-w", dest="write_fb", type=int, default=2)
    p.add_argument("-b", dest="nbit", type=int, default=2)
    p.add_argument("-P", dest="npol", type=int, default=1)
    p.add_argument("-r", dest="rfi_mode", type=int, default=2)
    p.add_argument("-s", dest="single_pass", action="store_true")
    p.add_argument("-t", dest="profile_pass", action="store_true")
    p.add_argument("-g", dest="gpu_id", type=int, default=0)
    p.add_argument("-p", dest="legacy_p", default=None)      # ignored
    # --- extensions (not in the reference) ---
    p.add_argument("--replay", nargs="+", default=None, help="dump file(s) instead of ring -k")
    p.add_argument("--datadir", default=sigproc.DATADIR)
    p.add_argument("--logdir", default=LOGDIR)
    p.add_argument("--out-sink", default=None, help="file standing in for ring -K")
    p.add_argument("--co-sink", default=None, help="file standing in for ring -C")
    p.add_argument("--no-control", action="store_true", help="do not join the multicast control group")
    p.add_argument("--fft-backend", choices=["lds", "hipfft"], default="lds")
    p.add_argument("--taps", type=int, default=1)
    p.add_argument("--rows-per-seg", type=int, default=1024, help="test hook: shorter segments")
    return p


def validate(args):
    if args.nbit not in (2, 4, 8):
        raise SystemExit("Unsupported NBIT!")
    if args.rfi_mode not in (0, 1, 2):
        raise SystemExit("Unsupported RFI mode!")
    if args.npol not in (1, 2):
        raise SystemExit("Unsupported npol!")
    if args.rows_per_seg == 1024 and args.gpu_id not in (0, 1, 2, 3, 4, 5, 6, 7):
        raise SystemExit("Unsupported GPU id!")


class Log(object):
    """multilog stand-in: timestamped lines to the per-process log file and optionally stdout."""

    def __init__(self, logdir, to_stdout, suffix=""):
        self.fps = []
        stamp = time.strftime("%Y%m%d_%H%M%S", time.gmtime())
        # (suffix: ranks that share one process -- the threaded rehearsal of coadd_host.py -- get a file each)
        path = os.path.join(logdir, "%s_%s_process_%06d%s.log" % (stamp, socket.gethostname(), os.getpid(), suffix))
        try:
            os.makedirs(logdir, exist_ok=True)
            self.fps.append(open(path, "w"))
        except OSError:
            pass
        if to_stdout:
            self.fps.append(sys.stdout)

    def __call__(self, level, msg):
        line = "[%s] %s%s" % (time.strftime("%Y-%m-%d-%H:%M:%S", time.gmtime()),
                              "ERR: " if level == "ERR" else "", msg)
        for fp in self.fps:
            fp.write(line if line.endswith("\n") else line + "\n")
            fp.flush()


def _pinned_bytes(n):
    """n bytes of page-locked host memory as a numpy array (plain numpy if torch cannot provide it)."""
    try:
        import torch
        t = torch.empty(n, dtype=torch.uint8, pin_memory=True)
        a = t.numpy()                   # shares the tensor's storage and keeps it alive
        return a
    except Exception:
        return np.empty(n, np.uint8)



def open_control_socket():
    """Non-blocking membership of 224.3.29.71:20000 (src/utils.c:619, process_baseband.cu:764)."""
    try:
        s = socket.socket(socket.AF_INET, socket.SOCK_DGRAM, socket.IPPROTO_UDP)
        s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        s.bind(("", MC_READER_PORT))
        mreq = struct.pack("4sl", socket.inet_aton(MC_GROUP), socket.INADDR_ANY)
        s.setsockopt(socket.IPPROTO_IP, socket.IP_ADD_MEMBERSHIP, mreq)
        s.setblocking(False)
        return s
    except OSError:
        return None


def test_for_cmd(sock, cmd):
    """src/utils.c:174-186: one non-blocking read of up to 32 bytes, any byte equal to cmd."""
    if sock is None:
        return False
    try:
        buf = sock.recv(32)
    except (BlockingIOError, OSError):
        return False
    return cmd in buf


def source_allowed(hdr, allow=None):
    """src/process_baseband.cu:893-913: position, then NAME, then DATAID (check_coords / check_name / check_id)"""
    allow = allow if allow is not None else load_write_allow()
    names, ids = allow[0], allow[1]
    coords = allow[2] if len(allow) > 2 else ()
    try:
        ra, dec = float(hdr.get("RA", 0)), float(hdr.get("DEC", 0))
    except ValueError:
        ra = dec = 0.0
    if any(coord_dist(cra, ra, cde, dec) < COORD_TOL for cra, cde in coords):
        return True
    name = hdr.get("NAME", "")
    if any(n in name for n in names):
        return True
    return any(i in hdr.get("DATAID", "") for i in ids)


class SecondReader(object):
    """The frames of one observation, one second at a time (src/process_baseband.cu:1015-1067).

    Frames are placed by their own headers (thread id, frame number: :1017-1034), a second is closed by the
    first frame of another second (:1019, :1058) and handed on only then -- so the last second of an
    observation is dropped -- and a frame that never arrives leaves zeros.  The frames of one second are
    gathered in a page-locked block in bulk; when frames were dropped the block swallows the head of the next
    second, which is carried over, so that the stream re-aligns at once (the reference loses the dropped frame
    and nothing else).  Shared by process_baseband's loop and the coadder host (one reader per antenna)."""

    def __init__(self, ring, first, sec_bytes, log):
        self.ring, self.sec_bytes, self.log = ring, sec_bytes, log
        self.readinto = getattr(ring, "readinto", None)
        self.carry = bytes(first)          # frames of the second being assembled that have been read already
        self.current_sec = vdif.unpack_header(first)["second"]
        self.boundary = None

    def fill(self, block):
        """Assemble second `current_sec` in `block`.  -> (have, next_header): bytes of this second in the block
        (frames of other seconds behind them are marked invalid) and the header of the first frame of the next
        second; (None, None) at the end of data (the last, partial or whole, second is dropped)."""
        ring, sec_bytes, current_sec = self.ring, self.sec_bytes, self.current_sec
        have = len(self.carry)
        block[:have] = np.frombuffer(self.carry, np.uint8)
        self.carry = b""
        boundary = None    # bytes (>= one frame) of the next second, already read
        eod = False
        while boundary is None and not eod:
            if have < sec_bytes:
                if self.readinto is not None:
                    got = self.readinto(block[have:]) or 0
                else:
                    rest = ring.read(sec_bytes - have)
                    got = len(rest)
                    block[have:have + got] = np.frombuffer(rest, np.uint8)
                if got % vdif.VD_FRM:
                    self.log("INFO", "Packet size=%d, expected %d.  Aborting this observation."
                             % (got % vdif.VD_FRM, vdif.VD_FRM))
                    got -= got % vdif.VD_FRM
                    eod = True
                elif got < sec_bytes - have:
                    eod = True
                have += got
            else:
                nxt = ring.read(vdif.VD_FRM)     # block full and all of this second: the next frame decides
                if len(nxt) != vdif.VD_FRM:
                    if len(nxt):
                        self.log("INFO", "Packet size=%d, expected %d.  Aborting this observation." % (len(nxt), vdif.VD_FRM))
                    eod = True
                    break
                if vdif.unpack_header(nxt)["second"] != current_sec:
                    boundary = nxt
                continue                            # (a duplicate frame of this second: dropped)
            secs = block[:have].reshape(-1, vdif.VD_FRM)[:, :4].copy().view("<u4")[:, 0] & 0x3FFFFFFF
            other = np.nonzero(secs != current_sec)[0]
            if other.size:
                cut = int(other[0]) * vdif.VD_FRM
                boundary = block[cut:have].tobytes()
                # what follows the cut is not this second's: hide it from the device-side frame index
                block[cut:have].reshape(-1, vdif.VD_FRM)[:, 3] |= 0x80      # VDIF invalid-data bit
                have = cut
        self.boundary = boundary
        if boundary is None:
            return None, None
        return have, vdif.unpack_header(boundary[:vdif.VD_FRM])

    def seal(self, block, have):
        """frames were dropped: whatever an earlier second left in the tail of the block is not data"""
        if have < self.sec_bytes:
            block[have:].reshape(-1, vdif.VD_FRM)[:, 3] |= 0x80

    def advance(self):
        """the second in the block has been handed on: the next one starts with what was read beyond it"""
        self.current_sec = vdif.unpack_header(self.boundary[:vdif.VD_FRM])["second"]
        self.carry, self.boundary = self.boundary, None


def run(args, in_ring=None, out_ring=None, co_ring=None, handle=None, control_sock=None):
    """Process observations until the input ring is exhausted or CMD_QUIT.  Returns the exit
    status of the reference (0, or 1 after a >1 s data skip)."""
    validate(args)
    lp = importlib.import_module(_pkg + ".libpb")
    log = Log(args.logdir, args.stdout_output)
    log("INFO", "[PROCESS_BASEBAND] invoked with: \n%s" % " ".join(sys.argv))
    R = args.rows_per_seg
    frames_per_sec = R * SEG_PER_SEC * 12500 // vdif.VD_DAT     # per thread; 25600 at R = 1024
    sec_bytes = 2 * frames_per_sec * vdif.VD_FRM                # 257 638 400 at R = 1024
    if in_ring is None:
        in_ring = dada.FileRing(args.replay) if args.replay else dada.open_ring(args.key_in)
    if out_ring is None and (args.out_sink or args.key_out):
        out_ring = dada.FileSink(args.out_sink) if args.out_sink else dada.open_ring(args.key_out, "w")
    if co_ring is None and (args.co_sink or args.key_co):
        co_ring = dada.FileSink(args.co_sink) if args.co_sink else dada.open_ring(args.key_co, "w")
    own_handle = handle is None
    if handle is None:
        handle = lp.PbHandle(device=args.gpu_id, nant=1, nbit=args.nbit, npol=args.npol, rfi_mode=args.rfi_mode,
                             taps=args.taps, fft_backend=lp.FFT_LDS if args.fft_backend == "lds" else lp.FFT_HIPFFT,
                             rows_per_seg=R, max_seg=SEG_PER_SEC, inject_frb=args.inject_frb, nsets=2)
    trim = handle.trim
    nsets = handle.nsets
    ctl = control_sock if control_sock is not None else (None if args.no_control else open_control_socket())
    blocks = None          # page-locked staging for one second of frames, allocated once
    tsamp = 12500.0 / 128e6 * 8
    exit_status, quit_ = 0, False
    written_files = []

    while not quit_:
        if test_for_cmd(ctl, CMD_QUIT):
            break
        log("INFO", "Waiting for DADA header.")
        raw_hdr = in_ring.next_header()
        if raw_hdr is None:
            log("INFO", "Input ring closed.  Exiting.")
            break
        t_obs = time.time()
        hdr = vdif.ascii_header_parse(raw_hdr)
        log("INFO", "Beginning new observation.")
        handle.reset_history(0)          # taps=4: the FIR window does not span observations
        # One second of frames is assembled in page-locked memory (nsets + 1 buffers in turn: the H2D of
        # second k is known to be complete once its filterbank bytes have been fetched, which happens
        # before buffer k mod (nsets + 1) comes round again), read straight into place where the ring
        # supports it -- one copy from the ring instead of three.
        if blocks is None:
            blocks = [_pinned_bytes(sec_bytes) for _ in range(nsets + 1)]
        first = in_ring.read(vdif.VD_FRM)
        if len(first) != vdif.VD_FRM:
            log("ERR", "Problem reading first bloody frame!  Bailing.")
            return 1
        vh = vdif.unpack_header(first)
        if vh["frame"] != 0 and vh["thread"] != 0:      # sic: `&&` in the reference (:845)
            log("ERR", "Incoming data were not aligned!")
            return 1
        station = int(hdr.get("STATIONID", 0))
        t_unix = vdif.vdif_to_unixepoch(vh)
        fb, fb_kur, cofb, cofb_kur = sigproc.fb_names(t_unix, station, args.datadir)
        heimdall_file = fb_kur if args.rfi_mode else fb
        coheimdall_file = cofb_kur if args.rfi_mode else cofb
        write_to_null = args.write_fb == 0
        if args.write_fb == 1:
            if source_allowed(hdr, load_write_allow(log=log)):
                log("INFO", "Source %s matches target list, recording filterbank data." % hdr.get("NAME", ""))
            else:
                write_to_null = True
                log("INFO", "Source %s not on target list, disabling filterbank data." % hdr.get("NAME", ""))
        if write_to_null:
            log("INFO", "Filterbank output disabled.  Would have written to %s." % fb)
            fb = fb_kur = os.devnull
        fb_fp, fb_kur_fp = None, None
        if args.rfi_mode in (0, 2):
            fb_fp = open(fb, "wb")
            log("INFO", "Writing no-RFI-excision filterbanks to %s." % fb)
        else:
            fb_fp = open(fb_kur, "wb")
            log("INFO", "Writing RFI-excision filterbanks to %s." % fb_kur)
        if args.rfi_mode == 2:
            fb_kur_fp = open(fb_kur, "wb")
            log("INFO", "Writing RFI-excision filterbanks to %s." % fb_kur)
        mjd, mjd_sec = vdif.frame_mjd(vh), vdif.frame_mjd_sec(vh)
        if out_ring is not None:
            out_ring.write_header(vdif.ascii_header_format(sigproc.psrdada_out_header(
                hdr, vh, args.npol, args.nbit, heimdall_file, t_unix, mjd, mjd_sec)))
        if co_ring is not None:
            co_ring.write_header(vdif.ascii_header_format(sigproc.psrdada_out_header(
                hdr, vh, args.npol, args.nbit, coheimdall_file, t_unix, mjd, mjd_sec)))
        sp_hdr = sigproc.sigproc_header(station, float(hdr.get("RA", 0)), float(hdr.get("DEC", 0)),
                                        hdr.get("NAME", ""), vdif.frame_dmjd(vh, frames_per_sec), args.npol, args.nbit)
        fb_fp.write(sp_hdr)
        if fb_kur_fp:
            fb_kur_fp.write(sp_hdr)

        current_sec = vh["second"]
        log("INFO", "Starting sec=%d, thread=%d" % (current_sec, vh["thread"]))
        st = dict(integrated_sec=0, fb_bytes=0, t_rt=time.time())
        out_buf = []           # 10-s buffer of the stream heimdall gets (:691-697)
        prof = dict(read=0.0, todev=0.0, write=0.0)
        if args.profile_pass:
            handle.timers(reset=True)
            handle.profile(True)

        def collect(k):
            """filterbank bytes of queued second k -> files and rings (:1364-1441, :1482-1494)"""
            t0 = time.time()
            handle.select_set(k % nsets)
            out = handle.fetch(0, 0, SEG_PER_SEC, raw=args.rfi_mode != 1, kur=args.rfi_mode != 0)
            main_codes = out["raw"] if args.rfi_mode != 1 else out["kur"]
            heim_codes = out["kur"] if args.rfi_mode != 0 else out["raw"]
            for iseg in range(SEG_PER_SEC):
                sl = slice(iseg * trim, (iseg + 1) * trim)
                if co_ring is not None:
                    co_ring.write(heim_codes[sl])
                fb_fp.write(main_codes[sl].tobytes())
                if fb_kur_fp:
                    fb_kur_fp.write(out["kur"][sl].tobytes())
                st["fb_bytes"] += trim
            out_buf.append(heim_codes.copy())
            st["integrated_sec"] += 1
            if st["integrated_sec"] % 10 == 0:
                lag = (time.time() - st["t_rt"]) - 10.0 * (R / 1024.0)
                if lag > 0.5:
                    log("ERR", "Measured time exceeding integrated time: lag %.2f s" % lag)
                st["t_rt"] = time.time()
            if st["integrated_sec"] >= 10 and out_ring is not None:
                if st["integrated_sec"] == 10:
                    out_ring.write(np.concatenate(out_buf))      # the full 10-s buffer
                else:
                    out_ring.write(out_buf[-1])                  # then 1 s at a time
            if len(out_buf) > 10:
                out_buf.pop(0)
            prof["write"] += time.time() - t0

        # Seconds are assembled by a SecondReader (frames placed by their own headers, a second closed by the first
        # frame of the next one, dropped frames cost only themselves) and pipelined over the handle's buffer sets:
        # second k is queued (H2D + kernels) before the output of second k-1 is collected, so the device works
        # while the host reads and writes.
        reader = SecondReader(in_ring, first, sec_bytes, log)
        queued = 0             # seconds handed to the device
        while True:
            t0 = time.time()
            block = blocks[queued % len(blocks)]
            have, nh = reader.fill(block)
            current_sec = reader.current_sec
            prof["read"] += time.time() - t0
            if have is None:
                break                                   # end of data: the last (partial or whole) second is dropped
            if nh["second"] - current_sec > 1:
                log("ERR", "Major data skip!  (%d vs. %d; thread = %d) Aborting this observation."
                    % (nh["second"], current_sec, nh["thread"]))
                exit_status, quit_ = 1, True
                break
            if test_for_cmd(ctl, CMD_QUIT):
                log("INFO", "Received CMD_QUIT, indicating data taking is ceasing.  Exiting.")
                quit_ = True
                break
            reader.seal(block, have)
            inject_now = 1 if (args.inject_frb and current_sec % 60 == 0) else 0
            if inject_now:
                log("INFO", "Injecting an FRB with integrated = %.2f!!!." % float(queued))
            t0 = time.time()
            handle.select_set(queued % nsets)
            handle.submit_vdif(0, 0, block, second=current_sec, frame0=0)
            handle.process(SEG_PER_SEC, inject_now)
            prof["todev"] += time.time() - t0
            queued += 1
            if queued - st["integrated_sec"] >= nsets:
                collect(st["integrated_sec"])           # the oldest queued second, while the newest computes
            reader.advance()
        while st["integrated_sec"] < queued:
            collect(st["integrated_sec"])
        integrated_sec, fb_bytes = st["integrated_sec"], st["fb_bytes"]

        if out_ring is not None:
            out_ring.end_of_data()
        if co_ring is not None:
            co_ring.end_of_data()
        if hasattr(in_ring, "finish_observation"):
            in_ring.finish_observation()
        fb_fp.close()
        if fb_kur_fp:
            fb_kur_fp.close()
        nsamp = fb_bytes * (8 // args.nbit) // 4096
        log("INFO", "Wrote %.2f MB (%.2f s) to %s" % (fb_bytes * 1e-6, nsamp * tsamp, fb))
        log("INFO", "Proc Time...%.3f" % (time.time() - t_obs))
        if args.profile_pass:
            # the reference's PROFILE block (:1538-1556).  Its kernels are fused here: "Convert" lives in the
            # kurtosis and channeliser kernels, "FFT" is the channeliser (unpack + FFT + detect), and
            # normalise / pscrunch / tscrunch / digitise are one kernel, reported under "Normalize".
            handle.profile(False)
            tm = handle.timers(reset=True)
            ms = lambda k: tm.get(k, (0.0, 0))[0] * 1e-3
            log("INFO", "Read Time...%.3f" % prof["read"])
            log("INFO", "Copy To Dev.%.3f" % prof["todev"])
            log("INFO", "Kurtosis....%.3f" % ms("kurtosis"))
            log("INFO", "FFT.........%.3f" % (ms("channelize") + ms("fft") + ms("inject")))
            log("INFO", "Normalize...%.3f" % ms("detect"))
            log("INFO", "Write.......%.3f" % prof["write"])
        written_files.append((fb, fb_kur if args.rfi_mode == 2 else None))
        if args.profile_pass or args.single_pass:
            break
    if own_handle:
        handle.close()
    run.last_files = written_files
    return exit_status


def main(argv=None):
    args = build_parser().parse_args(argv)
    sys.exit(run(args))


if __name__ == "__main__":
    main()
