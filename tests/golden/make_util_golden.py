#!/usr/bin/env python3
"""Golden vectors from the reference's own compiled code: oracle/_ref/libref_util.so = /root/reference/src/util.c
built as it lies (recipe: oracle/ref_util.mk; no stand-ins).  Run in the build container only:

    make -C oracle -f ref_util.mk && python tests/golden/make_util_golden.py

Writes tests/golden/reference_util.json:
  * "headers": for each case the exact bytes the reference's send_string / send_int / send_double (util.c:51-82)
    emit when called in the key sequence of write_sigproc_header (src/process_baseband.cu:243-268), whole and per
    key.  The VALUES handed to them (src_raj / src_dej in C float, fch1, foff, tsamp ...) are evaluated here the way
    :231-259 spell them (numpy float32 for the C floats): process_baseband.cu itself cannot be compiled in this
    image (nvcc / psrdada / vdifio absent), so the values are this script's reading of it, the ENCODING is the
    reference's own machine code.
  * "send": a few isolated send_* calls (empty string, negative int, NaN-free doubles) for the encoder alone.
  * "check_name" / "check_id" / "check_coords": the reference's answers (util.c:91-152) for a list of inputs.
The reference's text is not stored: only inputs and outputs.
"""
import ctypes as C
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(ROOT, "oracle", "_ref", "libref_util.so")

NFFT, NCHAN, NSCRUNCH, VLITE_RATE, CHANMIN, CHANMAX = 12500, 6251, 8, 128000000, 2155, 6250

CASES = [  # station, ra [rad], dec [rad], name, tstart (DMJD), npol, nbit
    dict(station=7, ra=0.8718, dec=-0.72452, name="B0833-45", tstart=57570 + 3600 / 86400., npol=1, nbit=8),
    dict(station=12, ra=1.4596726, dec=0.3842247, name="B0531+21", tstart=58849.123456789, npol=2, nbit=2),
    dict(station=99, ra=5.0715, dec=0.19, name="J1922+1053_coadd", tstart=59000.0, npol=1, nbit=4),
    dict(station=0, ra=0.0, dec=0.0, name="", tstart=0.0, npol=1, nbit=2),
    dict(station=28, ra=6.283185, dec=-1.5707, name="R3", tstart=58000.999988425926, npol=1, nbit=8),
]

NAMES = ["B0329+54", "PSR_B0329+54_x", "J0332+54", "B0531+21", "J0534+22", "B2319+60", "J2321+6024", "B0833-45",
         "J0835-45", "B1237+25", "B1933+16", "R2", "R3", "FRB121102_R1", "B0950+08", "B1133+16", "3C147", "3C48",
         "J0341+5711", "J1713+0747", "", "b0329+54", "CYGNUS-R2D2", "B0833", "VIRGO"]
IDS = ["18B-405", "19A-331", "SC1046", "VLASS1.1", "", "x19A-331y", "18B-40", "sc1046"]
COORDS = [(1.14479055, 1.28572588), (1.1448, 1.2857), (1.16, 1.2857), (0.5110324, 1.14737945), (0.52, 1.15),
          (4.755373, -0.344372), (4.76, -0.35), (4.7, -0.3), (0.0, 0.0), (1.14479055, 1.29572588), (1.14479055, 1.29573)]


def sigproc_angle(x):
    """src/process_baseband.cu:249-257: float hh = x; float mm = (hh-int(hh))*60; float ss = (mm-int(mm))*60;
    float out = int(hh)*1e4 + int(mm)*1e2 + ss  (the sum is evaluated in double, then narrowed to float)"""
    f = np.float32
    hh = f(x)
    mm = f(f(hh - f(int(hh))) * f(60))
    ss = f(f(mm - f(int(mm))) * f(60))
    return float(f(int(hh) * 1e4 + int(mm) * 1e2 + float(ss)))


def main():
    L = C.CDLL(LIB)
    libc = C.CDLL(None)
    libc.open_memstream.restype = C.c_void_p
    libc.open_memstream.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    libc.fclose.argtypes = [C.c_void_p]
    libc.free.argtypes = [C.c_void_p]
    send_string = L._Z11send_stringPKcP8_IO_FILE
    send_string.argtypes = [C.c_char_p, C.c_void_p]
    send_string.restype = None
    send_int = L._Z8send_intPKciP8_IO_FILE
    send_int.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
    send_int.restype = None
    send_double = L._Z11send_doublePKcdP8_IO_FILE
    send_double.argtypes = [C.c_char_p, C.c_double, C.c_void_p]
    send_double.restype = None
    check_name = L._Z10check_namePc
    check_name.argtypes = [C.c_char_p]
    check_id = L._Z8check_idPc
    check_id.argtypes = [C.c_char_p]
    check_coords = L._Z12check_coordsddd
    check_coords.argtypes = [C.c_double, C.c_double, C.c_double]

    def emit(calls):
        """bytes the reference writes for a list of (fn, args) calls, whole and per call"""
        whole, parts = b"", []
        for fn, args in calls:
            buf, size = C.c_void_p(), C.c_size_t()
            fp = libc.open_memstream(C.byref(buf), C.byref(size))
            fn(*(list(args) + [fp]))
            libc.fclose(fp)
            b = C.string_at(buf, size.value)
            libc.free(buf)
            parts.append(b)
            whole += b
        return whole, parts

    out = {"generated_from": "/root/reference/src/util.c via oracle/ref_util.mk", "headers": [], "send": []}
    chbw = -64. / NCHAN
    tsamp = float(NFFT) / VLITE_RATE * NSCRUNCH
    for c in CASES:
        raj = sigproc_angle((180 / math.pi) * (24. / 360) * c["ra"])
        dej = sigproc_angle((180 / math.pi) * math.fabs(c["dec"]))
        name = c["name"].encode()
        seq = [("HEADER_START", send_string, (b"HEADER_START",)), ("source_name", send_string, (b"source_name",)),
               ("source_name.value", send_string, (name,)), ("barycentric", send_int, (b"barycentric", 0)),
               ("telescope_id", send_int, (b"telescope_id", c["station"])), ("src_raj", send_double, (b"src_raj", raj)),
               ("src_dej", send_double, (b"src_dej", dej)), ("data_type", send_int, (b"data_type", 1)),
               ("fch1", send_double, (b"fch1", 384 + (CHANMIN - 0.5) * chbw)), ("foff", send_double, (b"foff", chbw)),
               ("nchans", send_int, (b"nchans", CHANMAX - CHANMIN + 1)), ("nbits", send_int, (b"nbits", c["nbit"])),
               ("tstart", send_double, (b"tstart", c["tstart"])), ("tsamp", send_double, (b"tsamp", tsamp)),
               ("nifs", send_int, (b"nifs", c["npol"])), ("HEADER_END", send_string, (b"HEADER_END",))]
        whole, parts = emit([(fn, a) for _, fn, a in seq])
        out["headers"].append(dict(c, src_raj=raj, src_dej=dej, bytes=whole.hex(),
                                   keys=[[k, p.hex()] for (k, _, _), p in zip(seq, parts)]))
    for kind, fn, args in (("string", send_string, (b"",)), ("string", send_string, (b"x" * 79,)),
                           ("int", send_int, (b"nbits", -1)), ("int", send_int, (b"telescope_id", 2147483647)),
                           ("double", send_double, (b"tstart", -0.0)), ("double", send_double, (b"foff", 1e-300)),
                           ("double", send_double, (b"src_dej", 123456.78125))):
        whole, _ = emit([(fn, args)])
        out["send"].append({"kind": kind, "args": [a.decode() if isinstance(a, bytes) else a for a in args],
                            "bytes": whole.hex()})
    out["check_name"] = [[n, int(bool(check_name(n.encode())))] for n in NAMES]
    out["check_id"] = [[n, int(bool(check_id(n.encode())))] for n in IDS]
    out["check_coords"] = [[ra, de, tol, int(bool(check_coords(ra, de, tol)))] for ra, de in COORDS for tol in (0.01, 0.001)]
    path = os.path.join(HERE, "reference_util.json")
    if "--out" in sys.argv:
        path = sys.argv[sys.argv.index("--out") + 1]
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, len(out["headers"]), "headers")


if __name__ == "__main__":
    main()
