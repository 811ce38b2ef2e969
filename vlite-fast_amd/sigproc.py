"""SIGPROC filterbank (.fil) headers and file names: the output contract of the hot path.

Byte format: /root/reference/src/util.c:51-82 (`send_string` = int32 length + bytes; ints as
native int32, doubles as native float64).  Key order and values:
/root/reference/src/process_baseband.cu:226-270 (`write_sigproc_header`), file names :272-304
(`get_fbfile`, `get_cofbfile`, `change_extension` :119-128).
"""
import math
import os
import struct
import time

import numpy as np

NFFT = 12500
NCHAN = NFFT // 2 + 1
NSCRUNCH = 8
VLITE_RATE = 128000000
CHANMIN = 2155
CHANMAX = 6250
DATADIR = "/mnt/ssd/fildata"      # src/def.h:28


def send_string(s):
    b = s.encode("ascii") if isinstance(s, str) else bytes(s)
    return struct.pack("=i", len(b)) + b


def send_int(name, v):
    return send_string(name) + struct.pack("=i", int(v))


def send_double(name, v):
    return send_string(name) + struct.pack("=d", float(v))


def _sigproc_angle(x_deg_or_hours):
    """`float hh = ...; float mm = (hh-int(hh))*60; float ss = (mm-int(mm))*60;
    float out = int(hh)*1e4 + int(mm)*1e2 + ss;`  -- all intermediates are C floats."""
    f = np.float32
    hh = f(x_deg_or_hours)
    mm = f(f(hh - f(int(hh))) * f(60))
    ss = f(f(mm - f(int(mm))) * f(60))
    return f(int(hh) * 1e4 + int(mm) * 1e2 + float(ss))


def sigproc_header(station_id, ra, dec, name, dmjd, npol, nbit):
    """Bytes of the header write_sigproc_header emits.  ra/dec in radians (VLA convention);
    the reference computes src_raj/src_dej in float and drops the sign of the declination
    (src/process_baseband.cu:249-259)."""
    chbw = -64. / NCHAN
    tsamp = float(NFFT) / VLITE_RATE * NSCRUNCH
    out = [send_string("HEADER_START"), send_string("source_name"), send_string(name),
           send_int("barycentric", 0), send_int("telescope_id", station_id)]
    raj = _sigproc_angle((180 / math.pi) * (24. / 360) * ra)
    out.append(send_double("src_raj", float(raj)))
    dej = _sigproc_angle((180 / math.pi) * math.fabs(dec))
    out.append(send_double("src_dej", float(dej)))
    out += [send_int("data_type", 1),
            send_double("fch1", 384 + (CHANMIN - 0.5) * chbw),
            send_double("foff", chbw),
            send_int("nchans", CHANMAX - CHANMIN + 1),
            send_int("nbits", nbit),
            send_double("tstart", dmjd),
            send_double("tsamp", tsamp),
            send_int("nifs", npol),
            send_string("HEADER_END")]
    return b"".join(out)


def read_header(buf):
    """Parse a SIGPROC header; returns (dict, header_length)."""
    ints = {"barycentric", "telescope_id", "data_type", "nchans", "nbits", "nifs", "machine_id", "nbeams", "ibeam"}
    dbls = {"src_raj", "src_dej", "fch1", "foff", "tstart", "tsamp", "az_start", "za_start", "refdm", "period"}
    pos = 0

    def rstr():
        nonlocal pos
        (n,) = struct.unpack_from("=i", buf, pos)
        pos += 4
        s = bytes(buf[pos:pos + n]).decode("ascii")
        pos += n
        return s

    hdr = {}
    assert rstr() == "HEADER_START"
    while True:
        key = rstr()
        if key == "HEADER_END":
            break
        if key in ints:
            (hdr[key],) = struct.unpack_from("=i", buf, pos)
            pos += 4
        elif key in dbls:
            (hdr[key],) = struct.unpack_from("=d", buf, pos)
            pos += 8
        else:
            hdr[key] = rstr()
    return hdr, pos


def fb_names(unix_epoch_seconds, station_id, datadir=DATADIR):
    """(fbfile, fbfile_kur, cofbfile, cofbfile_kur) as get_fbfile / get_cofbfile /
    change_extension build them; CHANMIN < 2411 gives the `_muos_` infix, the coadd variant is
    station 99."""
    stamp = time.strftime("%Y%m%d_%H%M%S", time.gmtime(unix_epoch_seconds))[:15]
    infix = "_muos" if CHANMIN < 2411 else ""
    fb = os.path.join(datadir, "%s%s_ea%02d.fil" % (stamp, infix, station_id))
    co = os.path.join(datadir, "%s%s_ea%02d.fil" % (stamp, infix, 99))
    return fb, change_extension(fb, ".fil", "_kur.fil"), co, change_extension(co, ".fil", "_kur.fil")


def change_extension(name, oldext, newext):
    i = name.find(oldext)
    return name + newext if i < 0 else name[:i] + newext


def psrdada_out_header(inhdr, vdif_hdr, npol, nbit, fb_file, unix_epoch_seconds, mjd, mjd_sec):
    """Key/value pairs write_psrdada_header puts on the output rings
    (src/process_baseband.cu:136-201), in its order."""
    chbw = -64. / NCHAN
    tsamp = float(NFFT) / VLITE_RATE * NSCRUNCH * 1e6
    nchan = CHANMAX - CHANMIN + 1
    h = {}
    station = int(inhdr.get("STATIONID", 0))
    h["STATIONID"] = "%d" % station
    h["BEAM"] = "%d" % station
    h["RA"] = "%f" % float(inhdr.get("RA", 0))
    h["DEC"] = "%f" % float(inhdr.get("DEC", 0))
    h["NAME"] = inhdr.get("NAME", "")
    h["SCANSTART"] = "%f" % float(inhdr.get("SCANSTART", 0))
    h["NCHAN"] = "%d" % nchan
    h["BANDWIDTH"] = "%f" % (nchan * chbw)
    h["CFREQ"] = "%f" % (384. + 0.5 * (CHANMIN + CHANMAX - 1) * chbw)
    h["NPOL"] = "%d" % npol
    h["NBIT"] = "%d" % nbit
    h["TSAMP"] = "%f" % tsamp
    h["UTC_START"] = time.strftime("%Y-%m-%d-%H:%M:%S", time.gmtime(unix_epoch_seconds))
    h["UNIXEPOCH"] = "%f" % float(unix_epoch_seconds)
    h["VDIF_MJD"] = "%d" % mjd
    h["VDIF_SEC"] = "%d" % mjd_sec
    if fb_file:
        h["SIGPROC_FILE"] = fb_file
    return h
