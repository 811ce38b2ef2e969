#!/usr/bin/env python3
"""What does detect's presence cost the channeliser?  The headline channeliser (experiments build, PB_SKIP=2: detect's
launch left out -- results invalid, timing only) runs beside synthetic co-runners (tools/corun.hip) that each take ONE
kind of resource for a fixed time: LDS capacity only (sleeping workgroups that hold 57 KB: the slot a detect workgroup
takes), vector issue slots only, LDS bandwidth only, HBM reads only.  Printed per co-runner: the channeliser's
per-launch time (stage timer), ms per step, socket power and joules per step (amdsmi energy accumulator).
usage (GPU box):  make -C vlite-fast_amd/csrc exp && hipcc -O3 -shared -fPIC --offload-arch=gfx950 -o build/libcorun.so tools/corun.hip
                  PB_LIBPATH=$PWD/vlite-fast_amd/csrc/libpb_hip_exp.so PB_SKIP=2 python tools/corun_probe.py
                  PB_LIBPATH=$PWD/vlite-fast_amd/csrc/libpb_hip_exp.so PB_SKIP=0 python tools/corun_probe.py none   (the real pipeline)"""
import ctypes as C
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import synth_second

lp = importlib.import_module("vlite-fast_amd.libpb")
try:
    import amdsmi
    amdsmi.amdsmi_init()
    GPU = amdsmi.amdsmi_get_processor_handles()[0]
except Exception:
    amdsmi = None

co = C.CDLL(os.path.join(ROOT, "build", "libcorun.so"))
vp = C.c_void_p
co.corun_hold.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_double, vp]
co.corun_valu.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, vp]
co.corun_valu_lds.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, vp]
co.corun_ldsbw.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_double, vp]
co.corun_hbm.argtypes = [vp, C.c_int, C.c_int, vp, C.c_size_t, C.c_int, C.c_double, vp, vp]

dev = torch.device("cuda", 0)
S, NSETS = 10, 3
A = int(os.environ.get("CORUN_ANTS", "1"))
h = lp.PbHandle(device=0, nant=A, nbit=8, npol=1, rfi_mode=2, rows_per_seg=1024, max_seg=S, nsets=NSETS,
                taps=int(os.environ.get("CORUN_TAPS", "1")))
for a in range(A):
    sec = synth_second(torch, dev, 42 + a, h.seg_samples, S)
    torch.cuda.synchronize()
    for st in range(NSETS):
        h.select_set(st)
        for s in range(S):
            h.submit_planar_dev(a, s, sec[s][0].data_ptr(), sec[s][1].data_ptr(), h.seg_samples)
    h.sync()
    del sec
sink = torch.zeros(64, dtype=torch.float32, device=dev)
nbytes = torch.zeros(1, dtype=torch.int64, device=dev)
big = torch.empty(1 << 30, dtype=torch.uint8, device=dev)          # 1 GiB: beyond the 256-MB infinity cache
big.zero_()
sc = torch.cuda.Stream(device=dev)
torch.cuda.synchronize()


def energy():
    if amdsmi is None:
        return 0.0
    e = amdsmi.amdsmi_get_energy_count(GPU)
    return e["energy_accumulator"] * e["counter_resolution"] * 1e-6


def launch(kind, ms):
    s = sc.cuda_stream
    k = kind[0]
    if k == "none":
        return
    if k == "hold":
        rc = co.corun_hold(s, kind[1], kind[2], kind[3], ms, sink.data_ptr())
    elif k == "valu":
        rc = co.corun_valu(s, kind[1], kind[2], kind[3], kind[4], ms, sink.data_ptr())
    elif k == "valulds":      # valulds,nwg,threads,burst,idle,lds_bytes
        rc = co.corun_valu_lds(s, kind[1], kind[2], kind[3], kind[4], kind[5], ms, sink.data_ptr())
    elif k == "ldsbw":
        rc = co.corun_ldsbw(s, kind[1], kind[2], kind[3], ms, sink.data_ptr())
    elif k == "hbm":
        nbytes.zero_()
        torch.cuda.synchronize()
        rc = co.corun_hbm(s, kind[1], kind[2], big.data_ptr(), big.numel(), kind[3], ms, sink.data_ptr(), nbytes.data_ptr())
    assert rc == 0, (kind, rc)


def measure(kind, dur_ms=250.0):
    h.sync()
    torch.cuda.synchronize()
    h.profile(True)
    h.timers(reset=True)
    e0 = energy()
    launch(kind, dur_ms)
    time.sleep(0.002)
    t0, k = time.perf_counter(), 0
    while (time.perf_counter() - t0) * 1e3 < dur_ms - 25.0:
        h.select_set(k % NSETS)
        h.process(S)
        if k >= 2:
            h.select_set((k - 2) % NSETS)
            h.fetch_view(0, 1, S)
        k += 1
    h.sync()
    dt = time.perf_counter() - t0
    e1 = energy()
    tm = h.timers(reset=True)
    torch.cuda.synchronize()
    extra = ""
    if kind[0] == "hbm":
        extra = "  co-runner read %.2f TB/s" % (float(nbytes.item()) / (dur_ms * 1e-3) / 1e12)
    ch = tm["channelize"]
    dt_ = tm.get("detect", (0, 0))
    print("%-34s step %.4f ms  channelize %.4f ms/launch  detect %.4f  %.0f W  %.3f J/step%s"
          % (" ".join(str(x) for x in kind), dt / k * 1e3, ch[0] / max(1, ch[1]), dt_[0] / max(1, dt_[1]), (e1 - e0) / dt, (e1 - e0) / k, extra),
          flush=True)


KINDS = [("none",),
         ("hold", 256, 384, 57 * 1024), ("hold", 512, 384, 49 * 1024), ("hold", 128, 384, 57 * 1024), ("hold", 256, 64, 1024),
         ("hold", 768, 384, 57 * 1024),
         ("valu", 256, 384, 16, 26), ("valu", 256, 384, 16, 8), ("valu", 256, 384, 16, 0), ("valu", 1024, 256, 16, 0),
         ("ldsbw", 256, 384, 32), ("ldsbw", 256, 384, 4), ("ldsbw", 256, 384, 0),
         ("hbm", 256, 256, 16), ("hbm", 256, 256, 4), ("hbm", 1024, 256, 0),
         ("none",)]
if len(sys.argv) > 1:
    KINDS = [tuple(int(x) if x.lstrip("-").isdigit() else x for x in a.split(",")) for a in sys.argv[1:]]
from bench import kernel_source_hash
print("PB_SKIP=%s antennas=%d lib=%s kernel_source_sha16=%s" % (os.environ.get("PB_SKIP", "0"), A,
                                                              os.path.basename(os.environ.get("PB_LIBPATH", "shipped")), kernel_source_hash()))
DUR = float(os.environ.get("CORUN_MS", "250"))
measure(("none",), 150.0)      # warm-up
for kd in KINDS:
    measure(kd, DUR)
