#!/usr/bin/env python3
"""Timing experiment: does a buffer written by one kernel and read by the next stay in the 256-MB Infinity Cache?
Producer/consumer pairs over working sets of growing size: a fill kernel (write only) followed by a sum kernel (read
only) over the same buffer, and a copy (read + write).  GB/s by size."""
import time
import torch
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for mb in (16, 32, 64, 128, 192, 256, 384, 512, 1024, 2048):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, dtype=torch.float32, device=dev)
    b = torch.empty(n, dtype=torch.float32, device=dev)
    reps = max(4, 8192 // mb)
    for _ in range(3):
        a.fill_(1.0); s = a.sum()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    e[0].record()
    for _ in range(reps):
        a.fill_(2.0)
    e[1].record()
    for _ in range(reps):
        s = a.sum()
    e[2].record()
    for _ in range(reps):
        a.fill_(3.0); s = a.sum()
    e[3].record()
    for _ in range(reps):
        b.copy_(a)
    e[4].record()
    torch.cuda.synchronize()
    tw, tr, twr, tc = e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), e[2].elapsed_time(e[3]), e[3].elapsed_time(e[4])
    gb = mb / 1024.0
    print("%5d MB: write %7.0f GB/s  read %7.0f GB/s  write-then-read %7.0f GB/s (2x bytes)  copy %7.0f GB/s (2x bytes)"
          % (mb, gb * reps / tw * 1e3, gb * reps / tr * 1e3, 2 * gb * reps / twr * 1e3, 2 * gb * reps / tc * 1e3))
