"""VDIF frames and psrdada-style ASCII headers: the input contract of the hot path.

What the reference uses (paths under /root/reference):
  * 5032-byte frames = 32-byte VDIF header + 5000 8-bit samples, 25600 frames per second
    per polarisation ("thread"), src/process_baseband.h:15-18, src/def.h:15,23-24
  * header fields read by process_baseband through the third-party vdifio library
    (getVDIFThreadID / FrameNumber / FrameSecond / Epoch / FrameDMJD,
    src/process_baseband.cu:843-844,1001-1002,1017-1019,241); vdifio is not in the image, the
    bit layout is the VDIF 1.1 specification as decoded by analysis/baseband.py:17-28
  * the ring's 4096-byte ASCII header written by writer (src/writer.c:92-122) or genbase
    (src/genbase.cu:330-353) and read with ascii_header_get (src/process_baseband.cu:141-150)
"""
import calendar
import time

import numpy as np

VD_FRM = 5032
VD_DAT = 5000
VD_HDR = 32
FRAMESPERSEC = 25600
VLITE_RATE = 128000000
DADA_HDR_SIZE = 4096
DADA_TIMESTR = "%Y-%m-%d-%H:%M:%S"   # psrdada's DADA_TIMESTR


def pack_header(second, epoch, frame, station, threadid, frame_bytes=VD_FRM, nbit=8, invalid=False):
    """32-byte VDIF header as 8 little-endian uint32 words."""
    w = np.zeros(8, dtype="<u4")
    w[0] = (second & 0x3FFFFFFF) | (0x80000000 if invalid else 0)
    w[1] = (frame & 0xFFFFFF) | ((epoch & 0x3F) << 24)
    w[2] = (frame_bytes // 8) & 0xFFFFFF
    w[3] = (station & 0xFFFF) | ((threadid & 0x3FF) << 16) | (((nbit - 1) & 0x1F) << 26)
    return w


def unpack_header(buf):
    """Fields of one header (bytes-like or uint32 words).  Same decoding as the reference's
    VDIFHeader (analysis/baseband.py:19-28)."""
    w = np.frombuffer(bytes(buf[:VD_HDR]), dtype="<u4") if not isinstance(buf, np.ndarray) or buf.dtype != np.dtype("<u4") else buf
    threadid = int((w[3] >> 16) & 0x3FF)
    frame_length = int(w[2] & 0xFFFFFF) * 8
    return dict(second=int(w[0] & 0x3FFFFFFF), invalid=bool(w[0] >> 31), epoch=int((w[1] >> 24) & 0x3F),
                frame=int(w[1] & 0xFFFFFF), frame_length=frame_length, frame_nsamp=frame_length - VD_HDR,
                station=int(w[3] & 0xFFFF), threadid=threadid, thread=int(threadid != 0),
                nbit=int((w[3] >> 26) & 0x1F) + 1)


def epoch_unix(epoch):
    """Unix time of the start of a VDIF reference epoch (half-years since 2000-01-01)."""
    return calendar.timegm((2000 + epoch // 2, 1 + 6 * (epoch % 2), 1, 0, 0, 0, 0, 0, 0))


def vdif_to_unixepoch(hdr):
    """src/utils.c:498-514: mktime of the epoch start plus the seconds field, corrected for the
    local-time offset of mktime -- i.e. UTC."""
    return epoch_unix(hdr["epoch"]) + hdr["second"]


def epoch_for_unix(t):
    """(epoch, seconds) of a Unix time, as vdifio's setVDIFFrameTime chooses them."""
    tm = time.gmtime(t)
    epoch = (tm.tm_year - 2000) * 2 + (1 if tm.tm_mon >= 7 else 0)
    return epoch, int(t) - epoch_unix(epoch)


def frame_mjd(hdr):
    """getVDIFFrameMJD: integer MJD of the frame."""
    return 40587 + (epoch_unix(hdr["epoch"]) + hdr["second"]) // 86400


def frame_mjd_sec(hdr):
    """getVDIFFrameMJDSec: seconds into that day."""
    return (epoch_unix(hdr["epoch"]) + hdr["second"]) % 86400


def frame_dmjd(hdr, framepersec=FRAMESPERSEC):
    """getVDIFFrameDMJD(hdr, 25600) as used for the .fil tstart (src/process_baseband.cu:241)."""
    return float(frame_mjd(hdr)) + (float(frame_mjd_sec(hdr)) + float(hdr["frame"]) / float(framepersec)) / 86400.0


def frame_block(pol0, pol1, second, epoch, station, frame0=0, out=None):
    """Frame pol-planar samples the way genbase does (src/genbase.cu:445-486): per frame number
    first thread 0 then thread 1, 5000 samples each, the second rolling over every 25600 frames.
    pol0 / pol1: uint8, a multiple of 5000 long.  Returns the byte stream (uint8)."""
    pol0 = np.ascontiguousarray(pol0, np.uint8)
    pol1 = np.ascontiguousarray(pol1, np.uint8)
    assert pol0.size == pol1.size and pol0.size % VD_DAT == 0
    nfr = pol0.size // VD_DAT
    if out is None:
        out = np.empty((nfr, 2, VD_FRM), np.uint8)
    else:
        out = out.reshape(nfr, 2, VD_FRM)
    out[:, 0, VD_HDR:] = pol0.reshape(nfr, VD_DAT)
    out[:, 1, VD_HDR:] = pol1.reshape(nfr, VD_DAT)
    fnum = frame0 + np.arange(nfr, dtype=np.int64)
    secs = (second + fnum // FRAMESPERSEC).astype(np.uint32)
    fr = (fnum % FRAMESPERSEC).astype(np.uint32)
    hdr = np.zeros((nfr, 2, 8), dtype="<u4")
    hdr[:, :, 0] = (secs & 0x3FFFFFFF)[:, None]
    hdr[:, :, 1] = (fr | np.uint32((epoch & 0x3F) << 24))[:, None]
    hdr[:, :, 2] = VD_FRM // 8
    hdr[:, 0, 3] = (station & 0xFFFF) | (7 << 26)
    hdr[:, 1, 3] = (station & 0xFFFF) | (1 << 16) | (7 << 26)
    out[:, :, :VD_HDR] = hdr.view(np.uint8).reshape(nfr, 2, VD_HDR)
    return out.reshape(-1)


def deframe_block(block, second=None, frame0=0):
    """Host-side restatement of the reference's demux loop (src/process_baseband.cu:1015-1067)
    for one block: returns uint8 [2][nframes*5000] placed by (thread, frame number) relative to
    the block's time origin -- (second, frame0), or the first frame's header when second is None;
    missing frames stay zero, frames outside the block's span are ignored.  The product path does
    this on the GPU (pb_submit_vdif / pb_submit_vdif_at); this function serves tests and file tools."""
    block = np.asarray(block, np.uint8)
    nslots = block.size // VD_FRM
    fr = block[:nslots * VD_FRM].reshape(nslots, VD_FRM)
    w = fr[:, :VD_HDR].copy().view("<u4")
    sec = (w[:, 0] & 0x3FFFFFFF).astype(np.int64)
    num = (w[:, 1] & 0xFFFFFF).astype(np.int64)
    thr = (((w[:, 3] >> 16) & 0x3FF) != 0).astype(np.int64)
    sec0, num0 = (sec[0], num[0]) if second is None else (second, frame0)
    rel = (sec - sec0) * FRAMESPERSEC + num - num0
    nfr = nslots // 2
    out = np.zeros((2, nfr, VD_DAT), np.uint8)
    ok = (rel >= 0) & (rel < nfr) & ((w[:, 0] >> 31) == 0)
    out[thr[ok], rel[ok]] = fr[ok, VD_HDR:]
    return out.reshape(2, nfr * VD_DAT)


# ---------------------------------------------------------------------------
# psrdada ASCII headers ("KEY value" lines in a 4096-byte block)

def ascii_header_set(hdr, key, value):
    """hdr: dict preserving insertion order (the block's line order)."""
    hdr[key] = str(value)
    return hdr


def ascii_header_format(hdr, size=DADA_HDR_SIZE):
    txt = "".join("%-19s %s\n" % (k, v) for k, v in hdr.items())
    raw = txt.encode("ascii")
    if len(raw) >= size:
        raise ValueError("ASCII header does not fit in %d bytes" % size)
    return raw + b"\0" * (size - len(raw))


def ascii_header_parse(raw):
    """First whitespace-separated token after each key, like ascii_header_get with "%s"/"%d"/"%lf"."""
    if isinstance(raw, (bytes, bytearray, np.ndarray)):
        raw = bytes(raw).split(b"\0", 1)[0].decode("ascii", "replace")
    out = {}
    for line in raw.splitlines():
        parts = line.split()
        if len(parts) >= 2 and parts[0] not in out:
            out[parts[0]] = parts[1]
    return out


def writer_header(station, ra, dec, name, scanstart, dataid, epoch, second):
    """The ring header writer produces for each observation (src/writer.c:92-122)."""
    h = {}
    t = epoch_unix(epoch) + second
    ascii_header_set(h, "STATIONID", "%d" % station)
    ascii_header_set(h, "NCHAN", "1")
    ascii_header_set(h, "BANDWIDTH", "%f" % -64.)
    ascii_header_set(h, "CFREQ", "%f" % 352.)
    ascii_header_set(h, "NPOL", "2")
    ascii_header_set(h, "NBIT", "8")
    ascii_header_set(h, "TSAMP", "%f" % 0.0078125)
    ascii_header_set(h, "RA", "%f" % ra)
    ascii_header_set(h, "DEC", "%f" % dec)
    ascii_header_set(h, "NAME", name)
    ascii_header_set(h, "SCANSTART", "%f" % scanstart)
    ascii_header_set(h, "DATAID", dataid)
    ascii_header_set(h, "UTC_START", time.strftime(DADA_TIMESTR, time.gmtime(t)))
    ascii_header_set(h, "UNIX_TIMET", "%d" % t)
    return h
