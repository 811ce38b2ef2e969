#!/usr/bin/env python3
"""Play a genbase / voltage dump (4096-byte ASCII header + VDIF frames, the file form of ring 0x40) into a psrdada
ring through the shim of include/pb_dada.h -- the writer's place (src/writer.c) in tools/dada_selftest.sh.
usage: dump_to_ring.py <dump file> <key hex> [seconds per write = 1]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    dada = importlib.import_module("vlite-fast_amd.dada")
    path, key = sys.argv[1], int(sys.argv[2], 16)
    nsec = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    chunk = int(51200 * 5032 * nsec)
    ring = dada.open_ring(key, "w")
    with open(path, "rb") as f:
        ring.write_header(f.read(dada.DADA_HDR_SIZE))
        total = 0
        while True:
            b = f.read(chunk)
            if not b:
                break
            ring.write(np.frombuffer(b, np.uint8))
            total += len(b)
    ring.end_of_data()
    ring.close()
    print("wrote %d bytes (%.2f s of frames) to ring %x" % (total, total / (51200 * 5032.0), key))


if __name__ == "__main__":
    main()
