"""Deterministic synthetic inputs shared by tests/golden/make_golden.py and the tests.

Pure integer hashing (splitmix64 of a counter) so that the same bytes come out on any
NumPy version: the golden vectors in this directory were computed by the *reference*
on exactly these inputs, and the tests must regenerate them bit for bit.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed, n, offset=0):
    """n 64-bit words: splitmix64 finaliser of (seed*GOLDEN + offset + k)."""
    with np.errstate(over="ignore"):
        k = np.arange(offset, offset + n, dtype=np.uint64)
        z = (np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + k * np.uint64(0x9E3779B97F4A7C15) +
             np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def gauss(seed, n, offset=0):
    """Approximately N(0,1) float64 (sum of 8 uniform bytes, Irwin-Hall)."""
    w = splitmix64(seed, n, offset)
    s = np.zeros(n, dtype=np.int64)
    for b in range(8):
        s += ((w >> np.uint64(8 * b)) & np.uint64(0xFF)).astype(np.int64)
    return (s.astype(np.float64) - 1020.0) / 209.02152999153936   # sqrt(8*(256^2-1)/12)


def baseband_u8(seed, n, sigma=16.9, mean=128.5, offset=0):
    """genbase-like 8-bit voltages: Gaussian, mean 128.5, sigma 16.9 codes, clamped and
    truncated the way src/genbase.cu:689-708 digitises (trunc(x + 128.5), 0..255)."""
    g = gauss(seed, n, offset) * sigma + mean
    u = np.where(g <= 0, 0, np.where(g >= 255, 255, np.floor(g)))
    return u.astype(np.uint8)


def vdif_header_words(second, epoch, frame, station, threadid, frame_bytes=5032, nbit=8):
    """Eight little-endian uint32 words of a VDIF 1.1 header (no reference code involved:
    bit layout per the VDIF specification)."""
    w = np.zeros(8, dtype=np.uint32)
    w[0] = second & 0x3FFFFFFF
    w[1] = (frame & 0xFFFFFF) | ((epoch & 0x3F) << 24)
    w[2] = (frame_bytes // 8) & 0xFFFFFF
    w[3] = (station & 0xFFFF) | ((threadid & 0x3FF) << 16) | (((nbit - 1) & 0x1F) << 26)
    return w
