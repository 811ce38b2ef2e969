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

`open_ring(key)` returns a PsrdadaRing when a `psrdada` Python binding is importable on the
host (to be verified on a box that has psrdada), else raises.
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

    def next_header(self):
        if self._fp:
            self._fp.close()
            self._fp = None
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

    def readinto(self, arr):
        """Fill a uint8 numpy array (e.g. pinned staging) without an extra copy; returns the
        number of bytes read (short only at end of data)."""
        mv = memoryview(arr).cast("B")
        got = 0
        while got < len(mv):
            n = self._fp.readinto(mv[got:])
            if not n:
                break
            got += n
        return got

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


def open_ring(key, mode="r"):
    """psrdada ring by hexadecimal key (as -k/-K/-C pass it).  Needs a psrdada binding on the
    host; there is none in this image, so this fails loudly rather than pretending."""
    try:
        import psrdada  # noqa: F401
    except ImportError:
        raise RuntimeError("psrdada ring 0x%x requested but no psrdada binding is importable on this "
                           "host; use --replay FILE (FileRing) or run where psrdada is installed" % key)
    raise NotImplementedError("PsrdadaRing: to be wired and verified on a host with psrdada (SURVEY 8f-1)")
