#!/usr/bin/env python3
"""Energy experiment: joules per step of the pipeline (or of one of its kernels: PB_SKIP=1 leaves the channeliser
out, 2 detect -- results invalid, so the switch exists only in the experiments build: `make -C vlite-fast_amd/csrc exp`,
PB_LIBPATH=vlite-fast_amd/csrc/libpb_hip_exp.so) from amdsmi's energy accumulator, next to time and mean power.
Variant libraries (PB_LIBPATH) with parts of a kernel compiled out give the energy of those parts by difference.
usage: python tools/energy_probe.py [seconds] [taps]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import amdsmi
import torch
from bench import synth_second
lp = importlib.import_module("vlite-fast_amd.libpb")
amdsmi.amdsmi_init()
g = amdsmi.amdsmi_get_processor_handles()[0]
dur = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
taps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
S, NSETS = 10, 3
h = lp.PbHandle(device=0, nant=1, nbit=8, npol=1, rfi_mode=2, rows_per_seg=1024, max_seg=S, nsets=NSETS, taps=taps)
sec = synth_second(torch, dev, 42, h.seg_samples, S)
torch.cuda.synchronize()
for st in range(NSETS):
    h.select_set(st)
    for s in range(S):
        h.submit_planar_dev(0, s, sec[s][0].data_ptr(), sec[s][1].data_ptr(), h.seg_samples)
h.sync()

def run(d):
    t0, k = time.perf_counter(), 0
    while time.perf_counter() - t0 < d:
        h.select_set(k % NSETS); h.process(S)
        if k >= 2:
            h.select_set((k - 2) % NSETS); h.fetch_view(0, 1, S)
        k += 1
    h.sync()
    return k, time.perf_counter() - t0

def energy():
    e = amdsmi.amdsmi_get_energy_count(g)
    return e["energy_accumulator"] * e["counter_resolution"] * 1e-6      # joules

if os.environ.get("PB_SKIP") and "exp" not in os.path.basename(os.environ.get("PB_LIBPATH", "")):
    sys.exit("PB_SKIP is read by the experiments build only: make -C vlite-fast_amd/csrc exp; "
             "PB_LIBPATH=vlite-fast_amd/csrc/libpb_hip_exp.so")
run(0.5)
e0 = energy(); k, dt = run(dur); e1 = energy()
idle0 = energy(); time.sleep(1.0); idle1 = energy()
print("PB_SKIP=%s PB_LIBPATH=%s taps=%d: %.4f ms per step, %.4f J per step, %.0f W mean; idle %.0f W -> %.4f J per step above idle"
      % (os.environ.get("PB_SKIP", "0"), os.path.basename(os.environ.get("PB_LIBPATH", "shipped")), taps, dt / k * 1e3, (e1 - e0) / k, (e1 - e0) / dt,
         idle1 - idle0, ((e1 - e0) - (idle1 - idle0) * dt) / k))
