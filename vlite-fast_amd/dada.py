"""Ring-buffer endpoints of process_baseband: the psrdada HDU interface, abstracted.

The reference talks to three psrdada rings (/root/reference/src/process_baseband.cu:541-569):
  key_in  (-k, 0x40)  header block + byte stream of 5032-B VDIF frames   (read, ipcio_read)
  key_out (-K)        header + excised-stream codes, 10 s then 1 s writes (ipcio_write)
  key_co  (-C)        header + one segment of codes per write             (ipcio_write)
psrdada itself (SysV shared memory + semaphores, `ipcbuf_t` sync block) is third-party and
absent from this image, so its in-memory layout cannot be checked here.  The host therefore
programs against the small interface below; two implementations ship:

  MemoryRing   in-process (tests, and producer/consumer threads)
  FileRing     a file on disk: 4096-byte ASCII header followed by the frame stream -- the
               `readbase`-style replay of a recorded or generated dump
               (/root/reference/src/readbase.c, src/genbase.cu:294 `-e`).  The path may also be a
               FIFO: reads loop until the requested bytes have arrived, so a site with psrdada can
               pipe a live ring in (and the FileSink outputs back out) through a few lines of C
               around ipcio_read / ipcio_write -- INTEGRATION.md shows that bridge.

  PsrdadaRing  the real thing: ctypes over a flat C shim (include/pb_dada.h, csrc/pb_dada_shim.c) around the
               very psrdada calls the reference makes.  The shim is built where psrdada is installed;
               `open_ring(key)` returns a PsrdadaRing when it is, else raises with the build line.
"""
import os

import numpy as np

DADA_HDR_SIZE = 4096


class RingEOD(Exception):
    """End of data on the ring: the writer unlocked it (ends the observation)."""


class ReadRing(object):
    """What process_baseband needs from its input HDU."""

    def next_header(self):
        """Block until an observation header is available; None when the ring is closed."""
        raise NotImplementedError

    def read(self, nbytes):
        """ipcio_read: up to nbytes of the data stream; fewer (possibly 0) at end of data."""
        raise NotImplementedError


class WriteRing(object):
    def write_header(self, raw):
        raise NotImplementedError

    def write(self, buf):
        raise NotImplementedError

    def end_of_data(self):
        raise NotImplementedError


class MemoryRing(ReadRing, WriteRing):
    """One observation after another, held in memory."""

    def __init__(self):
        self._obs = []       # list of [header_bytes, bytearray, closed]
        self._ri = 0
        self._rpos = 0

    # writer side
    def write_header(self, raw):
        self._obs.append([bytes(raw), bytearray(), False])

    def write(self, buf):
        self._obs[-1][1] += bytes(memoryview(np.ascontiguousarray(buf)).cast("B")) if isinstance(buf, np.ndarray) else bytes(buf)

    def end_of_data(self):
        self._obs[-1][2] = True

    # reader side
    def next_header(self):
        if self._ri >= len(self._obs):
            return None
        self._rpos = 0
        return self._obs[self._ri][0]

    def read(self, nbytes):
        data = self._obs[self._ri][1]
        out = bytes(data[self._rpos:self._rpos + nbytes])
        self._rpos += len(out)
        if len(out) < nbytes:
            # end of this observation: the next next_header() moves on
            if self._rpos >= len(data):
                self._ri_done = True
        return out

    def finish_observation(self):
        self._ri += 1

    def observations(self):
        return [(h, bytes(d)) for h, d, _ in self._obs]


class FileRing(ReadRing):
    """Replay of dump files: each file = 4096-byte ASCII header + VDIF frame stream."""

    def __init__(self, paths):
        self._paths = [paths] if isinstance(paths, str) else list(paths)
        self._i = -1
        self._fp = None
        self._pool = None
        self._regular = None

    def next_header(self):
        if self._fp:
            self._fp.close()
            self._fp = None
        self._regular = None
        self._i += 1
        if self._i >= len(self._paths):
            return None
        self._fp = open(self._paths[self._i], "rb", buffering=0)
        hdr = self.read(DADA_HDR_SIZE)
        if len(hdr) != DADA_HDR_SIZE:
            raise IOError("%s: short header" % self._paths[self._i])
        return hdr

    def read(self, nbytes):
        """Up to nbytes; fewer only at end of data (a pipe may deliver a request in pieces)."""
        first = self._fp.read(nbytes)
        if first is None:
            first = b""
        if len(first) == nbytes or not first:
            return first
        parts, got = [first], len(first)
        while got < nbytes:
            more = self._fp.read(nbytes - got)
            if not more:
                break
            parts.append(more)
            got += len(more)
        return b"".join(parts)

    # a single thread copies ~9 GB/s out of the page cache: one second of frames (258 MB) would cost 28 ms, a
    # quarter of what the device needs for it; regular files are read by several threads at once
    _PAR_CHUNK = 8 << 20
    _PAR_THREADS = 8

    def readinto(self, arr):
        """Fill a uint8 numpy array (e.g. pinned staging) without an extra copy; returns the
        number of bytes read (short only at end of data)."""
        mv = memoryview(arr).cast("B")
        fd = self._fp.fileno()
        if len(mv) >= 2 * self._PAR_CHUNK and self._seekable():
            pos = os.lseek(fd, 0, os.SEEK_CUR)
            want = min(len(mv), max(0, os.fstat(fd).st_size - pos))
            if want >= 2 * self._PAR_CHUNK:
                if self._pool is None:
                    from concurrent.futures import ThreadPoolExecutor
                    self._pool = ThreadPoolExecutor(min(self._PAR_THREADS, os.cpu_count() or 1))

                def piece(a):
                    b, done = min(want, a + self._PAR_CHUNK), a
                    while done < b:
                        n = os.preadv(fd, [mv[done:b]], pos + done)
                        if n <= 0:
                            break
                        done += n
                    return done - a

                got = sum(self._pool.map(piece, range(0, want, self._PAR_CHUNK)))
                os.lseek(fd, pos + got, os.SEEK_SET)
                return got
        got = 0
        while got < len(mv):
            n = self._fp.readinto(mv[got:])
            if not n:
                break
            got += n
        return got

    def _seekable(self):
        if self._regular is None:
            import stat
            self._regular = stat.S_ISREG(os.fstat(self._fp.fileno()).st_mode)
        return self._regular

    def finish_observation(self):
        pass


class FileSink(WriteRing):
    """Output-ring stand-in that appends header and data to a file (for tests / offline use)."""

    def __init__(self, path):
        self._path = path
        self._fp = open(path, "wb")
        self.nwrites = []

    def write_header(self, raw):
        self._fp.write(bytes(raw))

    def write(self, buf):
        b = bytes(memoryview(np.ascontiguousarray(buf)).cast("B"))
        self._fp.write(b)
        self.nwrites.append(len(b))

    def end_of_data(self):
        self._fp.flush()

    def close(self):
        self._fp.close()


def write_dump(path, header_raw, stream):
    """Write a FileRing-compatible dump."""
    with open(path, "wb") as f:
        f.write(bytes(header_raw))
        f.write(bytes(memoryview(np.ascontiguousarray(stream)).cast("B")))


class PsrdadaRing(ReadRing, WriteRing):
    """A psrdada HDU through the flat C shim of include/pb_dada.h (vlite-fast_amd/csrc/pb_dada_shim.c,
    built where psrdada is installed: `make -C vlite-fast_amd/csrc dada PSRDADA=<prefix>`).  The calls and
    their order are the reference's (src/process_baseband.cu): connect :541-569; per observation
    lock_read + blocking header read + mark_cleared :799-832, ipcio_read :838/:1034, unlock_read :1513;
    lock_write + header + mark_filled(4096) :172-199, ipcio_write :1416-1422 / :1482-1494,
    unlock_write :1498-1511."""

    def __init__(self, key, mode="r", lib=None):
        import ctypes as C
        self._C = C
        self._L = lib if lib is not None else load_shim()
        self.key, self.mode = key, mode
        self._how = 0            # 0 nothing read yet in this observation, 1 ipcio_read, 2 block level (never mixed)
        self._threads = max(1, int(os.environ.get("PB_DADA_THREADS", "8")))
        err = C.create_string_buffer(256)
        self._d = self._L.pb_dada_open(C.c_uint32(key), 0 if mode == "r" else 1, err, C.c_uint64(len(err)))
        if not self._d:
            raise RuntimeError(err.value.decode() or "Unable to connect to PSRDADA buffer key=%x!" % key)

    def _pick(self):
        """how this observation's data are read, decided at its first read (psrdada does not let ipcio_read and the
        block-level interface alternate): 2 = block level (PB_DADA_THREADS > 1, the default), 1 = ipcio_read"""
        if self._how == 0:
            self._how = 2 if self._threads > 1 else 1
        return self._how

    # reader side
    def next_header(self):
        C = self._C
        buf = C.create_string_buffer(DADA_HDR_SIZE)
        n = self._L.pb_dada_next_header(self._d, buf)
        if n < 0:
            raise IOError("psrdada ring 0x%x: header read failed (%d)" % (self.key, n))
        if n == 0:
            return None                      # ring shut down (the reference then looks for CMD_QUIT, :810-823)
        return buf.raw

    def read(self, nbytes):
        C = self._C
        buf = C.create_string_buffer(nbytes)
        if self._pick() == 2:
            n = self._L.pb_dada_read_mt(self._d, buf, C.c_uint64(nbytes), 1)
        else:
            n = self._L.pb_dada_read(self._d, buf, C.c_uint64(nbytes))
        if n < 0:
            raise IOError("psrdada ring 0x%x: Error on nread=%d." % (self.key, n))
        return buf.raw[:n]

    def readinto(self, arr):
        """Data bytes straight into a uint8 numpy array (page-locked staging), no intermediate copy.  Reads of
        8 MiB or more are copied out of the ring's filled buffers by several threads (pb_dada_read_mt; one
        ipcio_read memcpy thread moves ~9 GB/s, 28 ms per second of data).  PB_DADA_THREADS=1 keeps the
        reference's ipcio_read for every read."""
        C = self._C
        if self._pick() == 2:
            n = self._L.pb_dada_read_mt(self._d, arr.ctypes.data_as(C.c_void_p), C.c_uint64(arr.size), self._threads)
        else:
            n = self._L.pb_dada_read(self._d, arr.ctypes.data_as(C.c_void_p), C.c_uint64(arr.size))
        if n < 0:
            raise IOError("psrdada ring 0x%x: Error on nread=%d." % (self.key, n))
        return int(n)

    def finish_observation(self):
        self._how = 0
        if self._L.pb_dada_end_read(self._d) < 0:
            raise IOError("psrdada ring 0x%x: dada_hdu_unlock_read failed" % self.key)

    # writer side
    def write_header(self, raw):
        raw = bytes(raw)
        if len(raw) != DADA_HDR_SIZE:
            raise ValueError("psrdada header blocks are %d bytes" % DADA_HDR_SIZE)
        if self._L.pb_dada_write_header(self._d, raw) < 0:
            raise IOError("psrdada ring 0x%x: header write failed" % self.key)

    def write(self, buf):
        a = np.ascontiguousarray(buf)
        n = self._L.pb_dada_write(self._d, a.ctypes.data_as(self._C.c_void_p), self._C.c_uint64(a.nbytes))
        if n != a.nbytes:
            raise IOError("psrdada ring 0x%x: ipcio_write wrote %d of %d bytes" % (self.key, n, a.nbytes))

    def end_of_data(self):
        if self._L.pb_dada_end_write(self._d) < 0:
            raise IOError("psrdada ring 0x%x: dada_hdu_unlock_write failed" % self.key)

    def close(self):
        if getattr(self, "_d", None):
            self._L.pb_dada_close(self._d)
            self._d = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_SHIM = None
SHIM_PATH = os.environ.get("PB_DADA_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libpb_dada.so")


def bind_shim(L):
    """declare the prototypes of include/pb_dada.h on a loaded library"""
    import ctypes as C
    vp = C.c_void_p
    L.pb_dada_open.restype = vp
    L.pb_dada_open.argtypes = [C.c_uint32, C.c_int, C.c_char_p, C.c_uint64]
    L.pb_dada_next_header.restype = C.c_int64
    L.pb_dada_next_header.argtypes = [vp, C.c_char_p]
    L.pb_dada_read.restype = C.c_int64
    L.pb_dada_read.argtypes = [vp, vp, C.c_uint64]
    L.pb_dada_read_mt.restype = C.c_int64
    L.pb_dada_read_mt.argtypes = [vp, vp, C.c_uint64, C.c_int]
    L.pb_dada_end_read.argtypes = [vp]
    L.pb_dada_write_header.argtypes = [vp, C.c_char_p]
    L.pb_dada_write.restype = C.c_int64
    L.pb_dada_write.argtypes = [vp, vp, C.c_uint64]
    L.pb_dada_end_write.argtypes = [vp]
    L.pb_dada_close.restype = None
    L.pb_dada_close.argtypes = [vp]
    return L


def load_shim(path=None):
    """dlopen the psrdada shim (libpb_dada.so).  It exists only where it was built against psrdada."""
    global _SHIM
    if path is None and _SHIM is not None:
        return _SHIM
    import ctypes as C
    p = path or SHIM_PATH
    if not os.path.exists(p):
        raise RuntimeError("psrdada ring requested but %s is not built: on a host with psrdada run "
                           "`make -C vlite-fast_amd/csrc dada PSRDADA=<psrdada prefix>` (or point PB_DADA_LIB at "
                           "the shim); without psrdada use --replay FILE / --out-sink / --co-sink" % p)
    L = bind_shim(C.CDLL(p))
    if path is None:
        _SHIM = L
    return L


def open_ring(key, mode="r"):
    """psrdada ring by key (as -k/-K/-C pass it, parsed as hexadecimal): a PsrdadaRing over the shim.
    Fails loudly when the shim has not been built (this image has no psrdada to build it against)."""
    return PsrdadaRing(key, mode)
