#!/usr/bin/env python3
"""Headline benchmark: baseband -> filterbank throughput of the HIP hot path on MI355X.

Contract:  python bench.py --gpus N --steps K --warmup W
  N = 1   one antenna, 128 MS/s dual-pol, RFI mode 2 (raw + excised streams), 8-bit out
          (BASELINE.json configs[1]).  A *step* = one second of that antenna
          (10 segments x 1024 FFT rows x 12500 samples x 2 pols = 256 MB of 8-bit voltages)
          through kurtosis flagging -> channeliser -> detect/scrunch/quantise, with the raw
          samples already resident in HBM and the filterbank bytes copied back to the host.
          The same JSON line carries sub-records measured right after the headline run:
            taps4   the same step with the 4-tap polyphase window (north_star's PFB);
            ingest  the same step with every second arriving from page-locked host memory as VDIF
                    frames through pb_submit_vdif (PCIe-inclusive; never `value`);
            search  one heimdall-sized gulp of the downstream dedispersion + boxcar search.
  N > 1   one rank per GPU: either a launcher has started them already (torch.distributed.run sets
          WORLD_SIZE, which must equal N) or `python bench.py --gpus N` starts them itself as child
          processes under torch.distributed.run before anything touches the GPU and relays rank 0's line and
          the exit code.  Fewer GPUs than ranks is an error (not a wrap onto one card).  Antennas shard one per
          GPU (weak scaling) and the per-step incoherent sum of the excised fp32 planes is an RCCL reduce to
          rank 0 over xGMI, which requantises the coadded second; the line then carries `rccl_ranks`, and at
          N = 8 a `configs3` sub-record (BASELINE configs[3]: 16 antennas, 2 per GPU).
Timing: W warm-up steps + K timed steps straight after set-up give `ms_per_step_cold`; then 100 untimed
steps, W warm-up steps and `--regions` (5) back-to-back regions of exactly K steps each, every one bracketed by
barrier + synchronise, MAX over ranks: `ms_per_step` / `value` are the MEDIAN region, min and max beside it.
Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import re
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NFFT, NCHANOUT, ROWS = 12500, 4096, 1024
HBM_PEAK_GBS = 8000.0           # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
SURVEY_UNFUSED_BYTES_PER_SEGMENT = 845_900_000      # SURVEY.md section 8(d), RFI mode 2: the reference's unfused kernel chain


def kernel_source_hash():
    """Identifies the kernels a committed PMC profile was taken with: sha256 over the kernel sources (k_*.hip and
    the headers they include; not the host side of the library, which moves no bytes on the device)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "vlite-fast_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if (f.startswith("k_") and f.endswith(".hip")) or f in ("fft_lds.h", "fft_consts.h", "pb_internal.h", "kurtosis_dev.h"):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


PRECONDITION_STEPS = 100      # untimed steps before the warm-up steps of every timed run (clock ramp, ~65 ms)


def synth_second(torch, dev, seed, seg_samples, nseg, rfi_frac=0.0):
    """genbase-style voltages on the GPU: Gaussian, mean 128.5, sigma 16.9 codes, clamped
    (src/genbase.cu:689-708 of the reference, whose default run has no RFI); optionally rfi_frac of
    the 500-sample blocks carry an impulsive, strongly non-Gaussian burst.  Even at rfi_frac = 0 the
    3-sigma D'Agostino test flags ~0.5 % of blocks, i.e. ~13 % of FFT rows take the excised-FFT path."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    out = []
    for s in range(nseg):
        pols = []
        for p in range(2):
            x = torch.randn(seg_samples, device=dev, generator=g) * 16.9 + 128.5
            nblk = seg_samples // 500
            bad = torch.rand(nblk, device=dev, generator=g) < rfi_frac
            burst = (torch.rand(seg_samples, device=dev, generator=g) - 0.5) * 180.0
            x = x + burst * bad.repeat_interleave(500)
            pols.append(x.clamp_(0, 255).to(torch.uint8))
        out.append(pols)
    return out


# ---------------------------------------------------------------------------------------------
# CPU baseline (reported only).  Runs BEFORE torch / HIP are touched: it forks a process pool.

_CPU_V = None


def _cpu_chunk(arg):
    p, i, n = arg
    import oracle as O
    rows = _CPU_V.shape[1] // NFFT
    lo, hi = rows * i // n, rows * (i + 1) // n
    O.filterbank(_CPU_V[p, lo * NFFT:hi * NFFT], nfft=NFFT)
    return hi - lo


def _cpu_pfb_chunk(arg):
    p, i, n = arg
    import oracle as O
    nwin = _CPU_V.shape[1] // (4 * NFFT) - 1            # windows of 4 x 12500 samples that yield spectra
    lo, hi = nwin * i // n, nwin * (i + 1) // n
    if hi > lo:                                          # windows [lo, hi) -> 4 (hi - lo) spectra, the same ones
        O.polyphase_filterbank(_CPU_V[p, lo * 4 * NFFT:(hi + 1) * 4 * NFFT], nchan=NFFT // 2, nwindow=4)
    return 4 * (hi - lo)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline():
    """The NumPy channeliser north_star names (analysis/baseband.py:filterbank, restated in oracle/oracle.py
    and pinned to the reference by tests/golden) on BASELINE config 0's input: 1 s of one antenna, dual-pol,
    from framed VDIF bytes (deframe + de-interleave + float32) to the 6251-channel detected plane, on this
    host's cores -- once on one core, once with the FFT rows split over every core this process may use."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import multiprocessing as mp
    import oracle as O
    rows = ROWS * 10
    rng = np.random.default_rng(42)
    nfr = rows * NFFT // 5000
    raw = np.zeros((2 * nfr, 5032), np.uint8)
    for i in range(0, 2 * nfr, 4096):       # bounded temporaries
        j = min(i + 4096, 2 * nfr)
        raw[i:j, 32:] = np.clip(rng.standard_normal((j - i, 5000), dtype=np.float32) * 16.9 + 128.5, 0, 255).astype(np.uint8)
    raw = raw.ravel()
    t0 = time.perf_counter()
    v = O.vdif_get_data(raw)                      # deframe + de-interleave + float32
    t1 = time.perf_counter()
    for p in range(2):
        O.filterbank(v[p], nfft=NFFT)
    t2 = time.perf_counter()
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    ncores = min(ncores, 16)                      # the GPU box gives one GPU's job a 16-core share of the host
    nsamp = rows * NFFT                           # dual-pol samples
    one = nsamp / (t2 - t0) / 1e6
    global _CPU_V
    _CPU_V = v
    chunks = [(p, i, ncores) for p in range(2) for i in range(ncores)]
    t3 = time.perf_counter()
    with mp.get_context("fork").Pool(ncores) as pool:
        pool.map(_cpu_chunk, chunks)
    t4 = time.perf_counter()
    allc = nsamp / ((t4 - t3) + (t1 - t0)) / 1e6   # the deframe pass is not parallelised
    # P2, the CPU baseline north_star names for the PFB mode: analysis/baseband.py:1207-1237 polyphase_filterbank
    # (restated in oracle/oracle.py, pinned to the imported reference by tests/golden) with nchan = 6250, nwindow = 4 --
    # the window the taps = 4 GPU path applies -- on the same deframed second: one core, then the windows split over
    # the pool.  It yields complex spectra (no detection), 4 per 50 000-sample window, 10 236 per pol and second.
    pfb = None
    try:
        t7 = time.perf_counter()
        nspec = sum(O.polyphase_filterbank(v[p], nchan=NFFT // 2, nwindow=4).shape[0] for p in range(2))
        t8 = time.perf_counter()
        with mp.get_context("fork").Pool(ncores) as pool:
            nspec_pool = sum(pool.map(_cpu_pfb_chunk, chunks))
        t9 = time.perf_counter()
        assert nspec_pool == nspec, (nspec_pool, nspec)
        pfb_one = nsamp / ((t8 - t7) + (t1 - t0)) / 1e6
        pfb_all = nsamp / ((t9 - t8) + (t1 - t0)) / 1e6
        pfb = {"value": round(pfb_all, 2), "unit": "Msamp/s", "cores": ncores, "kind": "port", "value_1core": round(pfb_one, 2),
               "x_realtime": round(pfb_all / 128.0, 4), "x_realtime_1core": round(pfb_one / 128.0, 4), "cpu_model": cpu_model(),
               "sample": "BASELINE configs[0]'s second (1.0 s of one antenna, dual-pol, deframed as above): NumPy "
                         "polyphase_filterbank(nchan=6250, nwindow=4) (analysis/baseband.py:1207-1237 restated), %d complex "
                         "spectra; 1 core (%.1f s), and the windows split over a %d-process pool (%.1f s)"
                         % (nspec, t8 - t7, ncores, t9 - t8)}
    except Exception as e:
        pfb = {"error": repr(e)[:200]}
    _CPU_V = None
    # ... and the hot path itself -- the oracle's C restatement of the reference's kernels K1-K11 (convertarray,
    # kurtosis, D'Agostino, apply, 12500-point FFT, detect/normalise, scrunch, requantise; oracle/pb_oracle.c) -- on four
    # 100-ms segments of the SAME workload as the GPU line (RFI mode 2, 8-bit out, 2 x 1024 rows of 12500 samples
    # each), one core: the scalar port that the parity tests use as the checker, timed as a baseline, not shipped.
    hot = None
    try:
        nseg_cpu = 4
        seg = np.clip(rng.standard_normal((2, ROWS * NFFT), dtype=np.float32) * 16.9 + 128.5, 0, 255).astype(np.uint8)
        bpr, bpk = np.zeros(2 * O.NCHAN, np.float32), np.zeros(2 * O.NCHAN, np.float32)
        t5 = time.perf_counter()
        for _ in range(nseg_cpu):
            O.segment(seg, ROWS, bpr, bpk, rfi_mode=2, npol=1, nbit=8)
        t6 = time.perf_counter()
        hot_rate = nseg_cpu * ROWS * NFFT / (t6 - t5) / 1e6
        hot = {"value": round(hot_rate, 2), "unit": "Msamp/s", "cores": 1, "kind": "port", "x_realtime": round(hot_rate / 128.0, 4),
               "sample": "BASELINE configs[1]'s step cut to %d of its 10 segments (%.1f s of CPU): the oracle's K1-K11 chain in C, "
                         "RFI mode 2, both output streams, 8-bit, one core" % (nseg_cpu, t6 - t5)}
    except Exception as e:      # (the baseline is a reported figure: never the reason a bench run fails)
        hot = {"error": repr(e)[:200]}
    return {"value": round(allc, 2), "unit": "Msamp/s", "cores": ncores, "kind": "port", "hot_path": hot, "pfb": pfb,
            "value_1core": round(one, 2), "x_realtime": round(allc / 128.0, 4), "cpu_model": cpu_model(),
            "sample": "BASELINE configs[0]: 1.0 s of one antenna, dual-pol, 51 200 VDIF frames: deframe + NumPy "
                      "|rfft(12500)|^2 over 20 480 rows (analysis/baseband.py:filterbank restated); 1 core, and the "
                      "rows split over a %d-process pool (this job's share of the host's cores)" % ncores}


def measured_traffic(stage, args, taps, A=1):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary of THIS build
    (tools/profile_gpu.sh + tools/summarise_profile.py; FETCH_SIZE / WRITE_SIZE collected in separate passes
    and corrected as MI355X_MICROARCH.md prescribes).  None when no summary matches the configuration or
    when the kernels have changed since the profile was taken (source hash recorded in the summary)."""
    import glob
    if args.backend != "lds" or args.rfi_mode != 2 or args.seg_per_step != 10 or A != 1 or args.rfi_frac:
        return None
    # (the library's rule, pb_fused_kurtosis: the rectangular-window channeliser flags its own rows unless
    #  PB_FUSE_KURTOSIS=0; taps = 4 always runs kurtosis pass + weights + k_channelize_pfb)
    fused = args.rfi_mode != 0 and taps == 1 and int(os.environ.get("PB_FUSE_KURTOSIS", "1") or 0) >= 1
    names = {"kurtosis": "k_kurtosis_row",
             "channelize": "k_channelize_pfb" if taps == 4 else ("k_channelize_kur" if fused else "k_channelize"),
             "detect": "k_detect2"}
    want = names.get(stage)
    best = None
    sha = kernel_source_hash()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("kernel_source_sha16") != sha:
            continue
        if ("taps=%d" % taps) not in d.get("bench_config", {}).get("workload", ""):
            continue
        for k, t in d.get("kernels", {}).items():
            if k == want or k.startswith(want + "<"):
                best = int(t["hbm_bytes_per_launch"])
    return best


def measured_valu(args, taps, A=1):
    """Vector instructions per launch of the step's kernels from the committed rocprofv3 PMC summary of THIS build
    (tools/profile_counters.sh + tools/summarise_counters.py: SQ_INSTS_VALU per kernel, kernel-source hash recorded).
    None when there is no summary of these kernels / this configuration."""
    import glob
    if args.backend != "lds" or args.rfi_mode != 2 or args.seg_per_step != 10 or A != 1 or args.rfi_frac:
        return None
    sha = kernel_source_hash()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_issue_counters.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("kernel_source_sha16") != sha or d.get("taps", 1) != taps:
            continue
        out = {k: int(v["SQ_INSTS_VALU"]) for k, v in d.get("kernels", {}).items() if "SQ_INSTS_VALU" in v and k != "k_copy_out"}
        if out:
            return {"per_launch": out, "packed_share": d.get("packed_share", {}), "source": os.path.basename(f)}
    return None


# cycles a SIMD needs per wave64 vector instruction with three or more waves to choose from (tools/ubench_valu.hip,
# profiles/r02_ubench_valu.txt: 1.90 for v_fma / v_add / v_mul_f32, 2.82 for the packed-f32 forms)
VALU_CYC_SCALAR, VALU_CYC_PACKED, N_SIMD = 1.90, 2.82, 1024


def valu_record(args, taps, ms_per_step, gfx_mhz):
    """The vector pipes' share of the step: instructions of the step's kernels x the measured cycles per instruction
    / (1024 SIMDs x the step's cycles at the clock the chip held).  The path is a streaming FFT: HBM is the
    roofline the metric names, this is the other ceiling beside it."""
    mv = measured_valu(args, taps)
    if mv is None:
        return None
    cyc = 0.0
    for k, n in mv["per_launch"].items():
        pk = float(mv["packed_share"].get(k, 0.5))
        cyc += n * (pk * VALU_CYC_PACKED + (1.0 - pk) * VALU_CYC_SCALAR)
    mhz = gfx_mhz or 2400.0
    step_cycles = ms_per_step * 1e-3 * mhz * 1e6
    return {"insts_per_step": int(sum(mv["per_launch"].values())), "per_kernel": mv["per_launch"],
            "packed_share": mv["packed_share"], "cycles_per_inst": {"scalar_f32": VALU_CYC_SCALAR, "packed_f32": VALU_CYC_PACKED},
            "simd_cycles_per_step": round(cyc / N_SIMD), "step_cycles": round(step_cycles), "gfx_mhz": round(mhz),
            "frac": round(cyc / N_SIMD / step_cycles, 4), "source": mv["source"]}


def _parse_corun(text):
    """rows of tools/corun_probe.py's output: {(PB_SKIP, co-runner): [(channelize ms per launch, W, J per step)]}, and
    the kernel-source hashes its header lines carry"""
    rows, skip, shas = {}, None, set()
    for line in text.splitlines():
        if line.startswith("PB_SKIP="):
            skip = line.split()[0].split("=")[1]
            m = re.search(r"kernel_source_sha16=([0-9a-f]+)", line)
            shas.add(m.group(1) if m else None)
        m = re.match(r"^(\S.*?)\s+step ([0-9.]+) ms\s+channelize ([0-9.]+) ms/launch\s+detect ([0-9.]+)\s+([0-9.]+) W\s+([0-9.]+) J/step", line)
        if m:
            rows.setdefault((skip, m.group(1).strip()), []).append((float(m.group(3)), float(m.group(5)), float(m.group(6))))
    return rows, shas


def residency_record():
    """What bounds the step: the channeliser's per-launch time alone, beside 256 SLEEPING workgroups that only hold the
    57 KB of LDS a detect workgroup takes (one per CU), and beside the real detect, with the socket power of each
    (tools/corun_probe.py + tools/corun.hip, profiles/r05_notes.md section 1).  MEASURED by this run when the experiments
    build is in the tree and at least as new as the kernel sources (`make -C vlite-fast_amd/csrc exp` +
    build/libcorun.so: the probe needs detect's launch left out, which the shipped library cannot do) -- the probe then
    runs as a child process for about a second of GPU time; otherwise QUOTED from the newest committed record taken
    with these kernel sources (hash check, like `traffic` and `valu`); None when neither exists."""
    import glob
    import subprocess
    sha = kernel_source_hash()
    csrc = os.path.join(ROOT, "vlite-fast_amd", "csrc")
    exp, corun = os.path.join(csrc, "libpb_hip_exp.so"), os.path.join(ROOT, "build", "libcorun.so")
    text, source, measured = None, None, False
    try:
        srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h"))]
        if os.path.exists(exp) and os.path.exists(corun) and os.path.getmtime(exp) >= max(os.path.getmtime(f) for f in srcs):
            out = []
            for skip, kinds in (("2", ["none", "hold,256,384,58368"]), ("0", ["none"])):
                env = dict(os.environ, PB_LIBPATH=exp, PB_SKIP=skip, CORUN_MS="150")
                p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "corun_probe.py")] + kinds, env=env,
                                   capture_output=True, text=True, timeout=180)
                if p.returncode != 0:
                    raise RuntimeError("corun_probe: " + p.stderr[-300:])
                out.append(p.stdout)
            text, source, measured = "\n".join(out), "measured by this run: tools/corun_probe.py on the experiments build", True
    except Exception as e:
        text, source = None, "in-run probe failed: %s" % repr(e)[:200]
    if text is None:
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_corun_probe*.txt")), reverse=True):
            t = open(f).read()
            if _parse_corun(t)[1] == {sha}:
                text, source = t, "profiles/%s (quoted: taken with these kernel sources, not re-measured by this run)" % os.path.basename(f)
                break
    if text is None:
        return None
    try:
        rows, _ = _parse_corun(text)
        alone = rows[("2", "none")][-1]
        held = rows[("2", "hold 256 384 58368")][-1]
        pipe = rows[("0", "none")][-1]
        return {"channelize_ms_per_launch": {"alone": alone[0], "beside_256_sleeping_57KB_workgroups": held[0], "beside_detect": pipe[0]},
                "socket_w": {"alone": alone[1], "beside_256_sleeping_57KB_workgroups": held[1], "beside_detect": pipe[1]},
                "measured_in_run": measured, "kernel_source_sha16": sha,
                "bound": "residency: three channeliser workgroups use 504 of 512 VGPRs per SIMD lane and 150 of 160 KB of LDS; a "
                         "detect workgroup per CU takes the place of one, at a socket power well below the cap",
                "source": source}
    except Exception as e:
        return {"error": repr(e)[:200]}


def power_record(torch, lp, args, dev, local, taps, seconds=2.0):
    """What the package does while the pipeline runs: socket power against its cap, the XCDs' clocks and the share of
    the time the firmware spent throttling for package power (amdsmi GPU metrics, ppt_residency_acc against
    accumulation_counter), over `seconds` of untimed steps.  Reported because the chip sits at its power cap while the
    pipeline runs; that costs the 10 - 12 % of clock it shows, while the step itself is bound by how many workgroups a
    CU holds (`roofline.residency`, profiles/r05_notes.md)."""
    try:
        import amdsmi
        import threading
        amdsmi.amdsmi_init()
        g = amdsmi.amdsmi_get_processor_handles()[local]
    except Exception as e:
        return {"error": "amdsmi: %s" % e}
    S = args.seg_per_step
    h = lp.PbHandle(device=local, nant=1, nbit=args.nbit, npol=1, rfi_mode=args.rfi_mode, rows_per_seg=ROWS, max_seg=S,
                    nsets=args.nsets, taps=taps, fft_backend=lp.FFT_LDS if args.backend == "lds" else lp.FFT_HIPFFT)
    n = h.seg_samples
    sec = synth_second(torch, dev, 42, n, S, rfi_frac=args.rfi_frac)
    torch.cuda.synchronize()
    for st in range(args.nsets):
        h.select_set(st)
        for s in range(S):
            h.submit_planar_dev(0, s, sec[s][0].data_ptr(), sec[s][1].data_ptr(), n)
    h.sync()
    del sec
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            try:
                m = amdsmi.amdsmi_get_gpu_metrics_info(g)
                clk = [c for c in m.get("current_gfxclks", []) if isinstance(c, (int, float)) and c > 0]
                samples.append((m.get("current_socket_power"), sum(clk) / len(clk) if clk else None, m.get("average_umc_activity")))
            except Exception:
                pass
            time.sleep(0.05)

    def steps(dur):
        t0, k = time.perf_counter(), 0
        while time.perf_counter() - t0 < dur:
            h.select_set(k % args.nsets)
            h.process(S)
            if k >= args.nsets - 1:
                h.select_set((k - args.nsets + 1) % args.nsets)
                h.fetch_view(0, 1 if args.rfi_mode else 0, S)
            k += 1
        h.sync()
        return k, time.perf_counter() - t0

    steps(0.3)                                    # (clock ramp)
    try:
        cap = amdsmi.amdsmi_get_power_cap_info(g).get("power_cap", 0) / 1e6
        m0 = amdsmi.amdsmi_get_gpu_metrics_info(g)
        th = threading.Thread(target=sampler, daemon=True)
        th.start()
        k, dt = steps(seconds)
        stop[0] = True
        th.join(timeout=5)
        m1 = amdsmi.amdsmi_get_gpu_metrics_info(g)
    except Exception as e:
        h.close()
        return {"error": "amdsmi: %s" % e}
    h.close()
    pw = [s[0] for s in samples if isinstance(s[0], (int, float))]
    ck = [s[1] for s in samples if s[1]]
    um = [s[2] for s in samples if isinstance(s[2], (int, float))]

    def delta(key):
        a, b = m0.get(key), m1.get(key)
        return (b - a) if isinstance(a, (int, float)) and isinstance(b, (int, float)) else None

    acc, ppt = delta("accumulation_counter"), delta("ppt_residency_acc")
    return {"socket_w": round(sum(pw) / len(pw), 1) if pw else None, "cap_w": cap or None,
            "frac_of_cap": round(sum(pw) / len(pw) / cap, 4) if pw and cap else None,
            "gfx_mhz": round(sum(ck) / len(ck)) if ck else None, "gfx_mhz_max": 2400,
            "power_throttle_residency": round(ppt / acc, 4) if acc and ppt is not None else None,
            "thermal_throttle_residency": round((delta("socket_thm_residency_acc") or 0) / acc, 4) if acc else None,
            "hbm_controller_activity_pct": round(sum(um) / len(um), 1) if um else None,
            "ms_per_step_while_sampled": round(dt / k * 1e3, 4), "seconds": seconds, "samples": len(pw),
            "note": "amdsmi GPU metrics while the pipeline runs untimed: socket power against the package cap, mean of the "
                    "XCDs' current clocks, ppt (package power) throttle residency = d ppt_residency_acc / d accumulation_counter"}


def algorithmic_bytes(args, h, n, world, taps):
    """compulsory HBM bytes per antenna-segment of each kernel"""
    nstreams = 2 if args.rfi_mode == 2 else 1
    pbytes = nstreams * 2 * ROWS * NCHANOUT * 4                # power planes per antenna-segment
    alg = {
        "kurtosis": 2 * n + h.nblk,
        "channelize": 2 * n + pbytes,
        "fft": nstreams * (2 * n * 4 + 2 * ROWS * 6251 * 8),
        "detect": (pbytes if args.backend == "lds" else nstreams * 2 * ROWS * 6251 * 8)
                  + nstreams * (h.trim + (h.ave_per_seg * 4 if world > 1 else 0)),
    }
    if args.backend == "hipfft":
        alg["kurtosis"] = 2 * n + nstreams * 2 * n * 4
    return alg


def run_chain(torch, dist, lp, args, dev, local, rank, world, taps, steps, warmup, ingest=False, coadd=None,
              ant_per_gpu=None, regions=None):
    """One measurement of the pipeline: a COLD timed region (the caller's W warm-up steps, then K timed steps, on a
    GPU that has just been set up: what the driver's literal command sees without preconditioning), then
    PRECONDITION_STEPS untimed steps + W warm-up steps and `regions` back-to-back timed regions of exactly K steps
    each (every one bracketed by barrier + synchronise on both sides, MAX over ranks).  ms_per_step = the median
    region; min / max are reported beside it.  ingest: every second arrives as VDIF frames from page-locked host
    memory (pb_submit_vdif), otherwise the samples are resident in HBM."""
    S = args.seg_per_step
    A = args.ant_per_gpu if ant_per_gpu is None else ant_per_gpu
    regions = args.regions if regions is None else regions
    # No cyclic-GC pass of the interpreter inside a timed region: a full collection over the objects torch's import
    # leaves behind takes 30-40 ms -- sixty steps' worth.  Collected HERE, before the set-up, and switched off until the
    # timed steps are done: a 40-ms pass between staging the input and the first step (where it sat until round 5) left
    # the GPU idle for longer than it keeps its clocks, so that the cold region measured a clock ramp the set-up's own
    # device work had already paid for.
    import gc
    gc.collect()
    gc.disable()
    if coadd is None:
        coadd = world > 1        # the incoherent sum (local sum -> reduce -> requantise on the root) is part of the step
    backend = lp.FFT_LDS if args.backend == "lds" else lp.FFT_HIPFFT
    NSETS = args.nsets   # >= 2: the D2H of second k and the host's collection of it overlap the kernels of the seconds
                         # after it (3: the second the host waits for was queued two steps ago)
    h = lp.PbHandle(device=local, nant=A, nbit=args.nbit, npol=1, rfi_mode=args.rfi_mode,
                    fft_backend=backend, rows_per_seg=ROWS, max_seg=S, keep_ave=coadd, nsets=NSETS, taps=taps)
    n = h.seg_samples
    blocks = None
    for a in range(A):
        sec = synth_second(torch, dev, 42 + rank * A + a, n, S, rfi_frac=args.rfi_frac)
        torch.cuda.synchronize()
        if ingest:
            # frame the second the way genbase / writer do and keep it in page-locked host memory
            vdif = importlib.import_module("vlite-fast_amd.vdif")
            p0 = torch.cat([sec[s][0] for s in range(S)]).cpu().numpy()
            p1 = torch.cat([sec[s][1] for s in range(S)]).cpu().numpy()
            nfr = p0.size // 5000
            blocks = [torch.empty(2 * nfr * 5032, dtype=torch.uint8, pin_memory=True) for _ in range(NSETS + 1)]
            vdif.frame_block(p0, p1, 3600, 33, 7, out=blocks[0].numpy())
            for b in blocks[1:]:
                b.copy_(blocks[0])
        else:
            for st in range(NSETS):           # the same synthetic second sits in both buffer sets
                h.select_set(st)
                for s in range(S):
                    h.submit_planar_dev(a, s, sec[s][0].data_ptr(), sec[s][1].data_ptr(), n)
            h.sync()
        del sec
    nant_total = world * A
    leg = None
    if coadd:
        # The incoherent sum (coadd.IncoherentCoadd, the class the coadder host runs) has a stream of its own: local
        # sum -> RCCL reduce -> requantise of batch k are ordered on it by the device and run beside the kernels of
        # batch k+1; no host synchronisation inside a step.  One antenna per GPU: detect writes the plane to be
        # reduced straight into the leg's buffer of the batch's set and the local sum needs no kernel.
        cmod = importlib.import_module("vlite-fast_amd.coadd")
        leg = cmod.IncoherentCoadd(h, nant_total, dev, root=0, backend=args.dist_backend, order=args.coadd_order,
                                   layout=args.coadd_layout, parts=int(os.environ.get("PB_COADD_PARTS", "7")))     # (parts: timing experiments)
        if world == 1 and getattr(args, "emulate_world", 0) > 1:
            leg.emulate_root_of(args.emulate_world, dev, layout="sliced" if args.coadd_layout != "root" else "root")
    nstream_out = (0, 1) if args.rfi_mode == 2 else ((0,) if args.rfi_mode == 0 else (1,))
    state = {"k": 0, "sink": 0, "coadds": 0}

    def collect(k):
        """filterbank bytes of batch k on the host (pinned mirror filled by the async D2H that
        follows detect, src/process_baseband.cu:1370-1375)"""
        h.select_set(k % NSETS)
        for a in range(A):
            for st in nstream_out:
                v = h.fetch_view(a, st, S)
                state["sink"] += int(v[0]) + int(v[-1])

    def finish_batch(j):
        """Batch j is done on the device once its filterbank bytes are here (collect waits for them), so its
        incoherent-sum leg -- local sum (nothing to launch with a coadd target), RCCL reduce, requantisation on
        the root -- is queued on the coadd stream without a device-side wait for detect, one step behind the
        batch itself, and runs beside the kernels of the batches after it."""
        collect(j)                                    # (selects buffer set j mod NSETS)
        if leg is not None and leg.parts:
            leg.queue(j % NSETS, S)
            if rank == 0 and leg.parts & 4 and state["coadds"]:
                v = leg.coadded(S, age=1)             # coadded bytes of the batch before
                state["sink"] += int(v[0])
            state["coadds"] += 1

    trace = [] if os.environ.get("PB_BENCH_TRACE") else None      # timing experiments: host time per step

    def step():
        k = state["k"]
        ta = time.perf_counter()
        h.select_set(k % NSETS)
        if ingest:
            h.submit_vdif(0, 0, blocks[k % len(blocks)].numpy(), second=3600, frame0=0)
        h.process(S)
        tb = time.perf_counter()
        if k >= NSETS - 1:
            finish_batch(k - (NSETS - 1))
        if trace is not None:
            trace.append((tb - ta, time.perf_counter() - tb))
        state["k"] = k + 1

    def drain():
        for kk in range(max(0, state["k"] - (NSETS - 1)), state["k"]):
            finish_batch(kk)
        state["k"] = 0

    # The GPU reaches its sustained clocks only after ~30 ms of work (20-step chunks from a cold start: 0.81, 0.66,
    # then 0.64 ms per step), which is longer than a short warm-up plus a short timed region last together.  Both
    # figures are reported: the COLD region (W warm-up steps, K timed steps, nothing before them) and the sustained
    # ones (PRECONDITION_STEPS untimed steps, W warm-up steps, then `regions` timed regions of K steps back to back).
    # (the interpreter's garbage collector is off since the top of this function)
    h.profile(True)      # (also while warming up: the stage timers' events are created once and then reused)

    def fence():
        drain()
        h.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    per_rank = []          # N > 1: every rank's own time of each timed region (the line reports min / max over ranks)

    def timed_region():
        """EXACTLY `steps` steps between two fences; seconds, MAX over ranks"""
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev if args.dist_backend == "nccl" else "cpu", dtype=torch.float64)
            allt = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(allt, t)
            per_rank.append([float(x.item()) for x in allt])
            dt = max(per_rank[-1])
        return dt

    for _ in range(warmup):
        step()
    fence()
    dt_cold = timed_region()
    for _ in range(PRECONDITION_STEPS + warmup):
        step()
    fence()
    h.timers(reset=True)
    h.profile(True)
    per_rank.clear()
    if leg is not None:
        leg.reduce_ms()
        leg.timing = True
    dts = [timed_region() for _ in range(max(1, regions))]
    red_ms, red_n = leg.reduce_ms() if leg is not None else (0.0, 0)
    gc.enable()
    dt = float(np.median(dts))
    if trace is not None:
        tr = np.array(trace[-steps:]) * 1e3
        print("host ms per step (taps %d%s): process mean %.3f max %.3f; collect mean %.3f max %.3f; slowest steps %s"
              % (taps, ", ingest" if ingest else "", tr[:, 0].mean(), tr[:, 0].max(), tr[:, 1].mean(), tr[:, 1].max(),
                 np.argsort(-tr.sum(axis=1))[:6].tolist()), file=sys.stderr)
    h.profile(False)
    tm = h.timers(reset=True)
    nsteps_timed = steps * len(dts)
    res = None
    if rank == 0:
        samples = float(nant_total) * S * n * steps          # dual-pol samples
        msamp = samples / dt / 1e6
        alg = algorithmic_bytes(args, h, n, world, taps)
        stages = {k: v for k, v in tm.items() if v[1] > 0 and k in alg}
        # the dominant kernel of an HBM roofline: the one with the most compulsory traffic per launch (it is also the
        # longest one alone; in the pipeline detect's launch -- first to last workgroup -- can span the whole step
        # although its workgroups only fill the slots the channeliser leaves, so wall time per launch does not rank)
        dom = max(stages, key=lambda k: alg[k])
        avg_ms = stages[dom][0] / stages[dom][1]
        per_launch = alg[dom] * S * A
        achieved = per_launch / (avg_ms * 1e-3) / 1e9
        res = {"msamp": msamp, "ms_per_step": dt / steps * 1e3, "nant_total": nant_total, "ant_per_gpu": A,
               "ms_per_step_cold": dt_cold / steps * 1e3,
               "regions": {"n": len(dts), "steps_each": steps, "ms_per_step_median": round(dt / steps * 1e3, 4),
                           "ms_per_step_min": round(min(dts) / steps * 1e3, 4),
                           "ms_per_step_max": round(max(dts) / steps * 1e3, 4)},
               "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1),
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                            "traffic": measured_traffic(dom, args, taps, A), "avg_launch_ms": round(avg_ms, 4),
                            "algorithmic_bytes_per_launch": per_launch},
               "stage_ms_per_step": {k: round(v[0] / nsteps_timed, 4) for k, v in tm.items() if v[1] > 0}}
        # the whole step against the same peak: compulsory bytes of every kernel that ran (the channeliser that flags
        # its own rows reads the voltages once: no separate kurtosis pass, 256 MB per second of data less than round 2)
        if world > 1:
            # the median region as every rank saw it, and the collective's device time on this (the root) rank: a curve
            # that bends can then be read (a slow rank? the reduce?)
            med = per_rank[int(np.argsort(dts)[len(dts) // 2])]
            res["per_rank"] = {"ms_per_step_min": round(min(med) / steps * 1e3, 4), "ms_per_step_max": round(max(med) / steps * 1e3, 4),
                               "ms_per_step": [round(x / steps * 1e3, 4) for x in med]}
            res["coadd_layout"] = leg.layout if leg is not None else None
            res["reduce_leg"] = {"device_ms_per_call_root": round(red_ms / red_n, 4) if red_n else None, "calls": red_n,
                                 "bytes_per_call": int(S * h.ave_per_seg * 4),
                                 "note": "event pair around the leg's collective(s) on its stream (sliced: all-to-all of plane "
                                         "slices + gather of code bytes, with this rank's tree and requantisation between "
                                         "them; root: gather of planes; fast: reduce), beside the next batch's kernels"}
        step_bytes = sum(alg[k] for k in stages) * S * A
        res["roofline"]["pipeline"] = {"algorithmic_bytes_per_step": step_bytes,
                                       "achieved": round(step_bytes / (dt / steps) / 1e9, 1),
                                       "frac": round(step_bytes / (dt / steps) / 1e9 / HBM_PEAK_GBS, 4)}
        # SURVEY.md 8(d)'s traffic model of the reference's UNFUSED kernel chain (17 kernels + cuFFT, every intermediate
        # through HBM): 845.9 MB per antenna-segment in RFI mode 2.  north_star's ">= 60 % HBM roofline" was written
        # against that model; read at this run's samples per second it asks for more than the peak, i.e. the fused
        # design does not move those bytes -- both readings are in the line.
        if args.rfi_mode == 2 and args.backend == "lds":
            sm = SURVEY_UNFUSED_BYTES_PER_SEGMENT * S * A
            res["roofline"]["survey_model"] = {"bytes_per_antenna_segment": SURVEY_UNFUSED_BYTES_PER_SEGMENT, "bytes_per_step": sm,
                                               "implied": round(sm / (dt / steps) / 1e9, 1), "unit": "GB/s",
                                               "frac": round(sm / (dt / steps) / 1e9 / HBM_PEAK_GBS, 4),
                                               "note": "SURVEY 8(d): HBM bytes the reference's unfused chain moves per antenna-segment "
                                                       "x this run's rate; frac > 1 means the fused kernels do not move those bytes "
                                                       "(they move algorithmic_bytes_per_step), not that the peak was exceeded"}
    if leg is not None:
        leg.close()
    h.close()
    return res


def alone_record(torch, lp, args, dev, local, taps, launches=10):
    """The kernels of one second back to back on ONE stream (one buffer set): hipEvent time per launch with
    nothing running beside them, and the dominant kernel's roofline fraction from that -- next to the figure
    of the timed region, where the previous second's detect and copy-out share the CUs."""
    S, A = args.seg_per_step, args.ant_per_gpu
    backend = lp.FFT_LDS if args.backend == "lds" else lp.FFT_HIPFFT
    h = lp.PbHandle(device=local, nant=A, nbit=args.nbit, npol=1, rfi_mode=args.rfi_mode, fft_backend=backend,
                    rows_per_seg=ROWS, max_seg=S, nsets=1, taps=taps)
    n = h.seg_samples
    for a in range(A):
        sec = synth_second(torch, dev, 42 + a, n, S, rfi_frac=args.rfi_frac)
        torch.cuda.synchronize()
        for s in range(S):
            h.submit_planar_dev(a, s, sec[s][0].data_ptr(), sec[s][1].data_ptr(), n)
        h.sync()
        del sec
    h.profile(True)
    for _ in range(60):               # (clock ramp after the set-up above, and the timers' events)
        h.process(S)
        h.sync()
    h.timers(reset=True)
    for _ in range(launches):
        h.process(S)
        h.sync()
    tm = h.timers(reset=True)
    alg = algorithmic_bytes(args, h, n, 1, taps)
    ms = {k: v[0] / v[1] for k, v in tm.items() if v[1] > 0}
    dom = max((k for k in ms if k in alg), key=lambda k: alg[k])
    per_launch = alg[dom] * S * A
    achieved = per_launch / (ms[dom] * 1e-3) / 1e9
    h.close()
    return {"kernel": dom, "avg_launch_ms": round(ms[dom], 4), "achieved": round(achieved, 1),
            "frac": round(achieved / HBM_PEAK_GBS, 4), "launches": launches,
            "ms_per_launch": {k: round(v, 4) for k, v in ms.items()}}


def search_record(lp, torch):
    """Downstream search (BASELINE config 5) on one heimdall-sized gulp, production flags of
    scripts/start_heimdall_single_antenna:21: 30 720 samples x 4096 channels of 8-bit codes, DM 2-1000 in
    steps of 2 (500 trial DMs), boxcars 2^0..2^6, zapped channel ranges, 2-s running baseline; codes from
    page-locked host memory (H2D included), the above-threshold (DM, sample) list back.  Real-time factor =
    new samples per gulp (gulp minus the largest delay) / time."""
    search = importlib.import_module("vlite-fast_amd.search")
    T = search.HEIMDALL_GULP
    rng = np.random.default_rng(1)
    pinned = torch.empty((T, NCHANOUT), dtype=torch.uint8, pin_memory=True)
    codes = pinned.numpy()
    codes[:] = np.clip(rng.normal(127.5, 1 / 0.02957, (T, NCHANOUT)), 0, 255).astype(np.uint8)
    with search.Searcher(max_samples=T, dm_step=2.0) as s:
        s.set_baseline(2560)
        s.peaks(codes, 6.0)
        n, t0 = 5, time.perf_counter()
        for _ in range(n):
            s.peaks(codes, 6.0)
        dt = (time.perf_counter() - t0) / n
        tout = T - s.max_delay
        return {"ms_per_gulp": round(dt * 1e3, 3), "new_seconds_per_gulp": round(tout * s.tsamp, 3),
                "x_realtime": round(tout * s.tsamp / dt, 1), "ndm": int(s.ndm), "nboxcar": int(s.nbox),
                "nsamps_gulp": T, "stage_ms": {k: round(v, 3) for k, v in s.timers().items()}}


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--regions", type=int, default=5,
                    help="back-to-back timed regions of --steps steps each (median reported, min / max beside it)")
    ap.add_argument("--backend", choices=["lds", "hipfft"], default="lds")
    ap.add_argument("--rfi-mode", type=int, default=2)
    ap.add_argument("--nbit", type=int, default=8)
    ap.add_argument("--seg-per-step", type=int, default=10)
    ap.add_argument("--ant-per-gpu", type=int, default=1)
    ap.add_argument("--rfi-frac", type=float, default=0.0,
                    help="fraction of 500-sample blocks given an impulsive RFI burst (default: clean noise)")
    ap.add_argument("--taps", type=int, default=1, help="1 = rectangular window (reference GPU path), 4 = PFB")
    ap.add_argument("--nsets", type=int, default=3,
                    help="buffer sets (1 = no batch pipelining; 3 = the host collects batch k - 2 after queuing batch "
                         "k, so that its wait for a copy-out never keeps the next batch from being queued)")
    ap.add_argument("--dist-backend", default="nccl",
                    help="nccl (= RCCL, default); rehearsals: gloo (one process per rank) or threads (the ranks as threads "
                         "of this process, with --share-gpus: for boxes that allow few processes on a card)")
    ap.add_argument("--coadd-order", choices=["tree", "fast"], default="tree",
                    help="N > 1: tree (default) = the defined order of the fp32 additions (local tree, gather to rank 0, "
                         "root tree: the same coadded bytes on any number of GPUs); fast = one RCCL reduce(SUM)")
    ap.add_argument("--coadd-layout", choices=["auto", "root", "sliced"], default="auto",
                    help="N > 1, tree order: sliced (auto) = all-to-all of plane slices, every rank sums and requantises 1/N "
                         "of the plane, code bytes gathered to rank 0; root = every plane gathered to rank 0")
    ap.add_argument("--share-gpus", action="store_true",
                    help="rehearsal only: let more ranks than there are GPUs run (ranks wrap onto the cards); "
                         "without it a run with fewer GPUs than ranks fails")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port of the ranks --gpus N starts (0: pick one)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-power", action="store_true", help="skip the 2-s power / clock / throttle sample (roofline.power)")
    ap.add_argument("--no-residency", action="store_true",
                    help="skip roofline.residency (its in-run form starts a child process: not wanted under a profiler)")
    ap.add_argument("--no-extras", action="store_true", help="skip the taps4 / ingest / search / configs3 sub-records")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="with --coadd-selftest: also run the ROOT's tree over this many gathered planes per step (timing of "
                         "rank 0's extra device work in a world of that size; no collective, coadded bytes invalid)")
    ap.add_argument("--coadd-selftest", action="store_true",
                    help="N = 1 only: print the step with the incoherent-sum leg of the N > 1 path switched on "
                         "(fp32 planes kept, local sum, RCCL reduce in a one-rank group, requantisation) instead of the bench line")
    return ap


def launcher_argv(args, argv, port):
    """Command line of the N ranks `bench.py --gpus N` starts when no launcher has started them already: one
    process per GPU under torch.distributed.run on this node, the way scripts/start_coadd:20-58 of the reference
    starts one coadder rank per antenna host under mpirun.  Rendezvous on 127.0.0.1."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def pick_json_line(text):
    """the bench line among whatever else the ranks printed: the last line that parses as a JSON object"""
    for line in reversed(text.splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                d = json.loads(line)
            except ValueError:
                continue
            if isinstance(d, dict):
                return line
    return None


def launch_ranks(args, argv, run=None):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: start the N ranks as CHILD
    processes (this process never touches the GPU and nothing is exec'd), relay rank 0's JSON line and the
    children's exit code."""
    import socket
    import subprocess
    port = args.master_port
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = launcher_argv(args, argv, port)
    p = (run or subprocess.run)(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = pick_json_line(p.stdout or "")
    if line is not None:
        print(line)
    else:
        sys.stdout.write(p.stdout or "")
    sys.stdout.flush()
    if p.returncode == 0 and line is None:
        print("bench.py: the ranks printed no JSON line", file=sys.stderr)
        return 1
    return p.returncode


def main():
    argv = sys.argv[1:]
    args = build_parser().parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.dist_backend == "threads" and args.gpus > 1:
        sys.exit(threaded_main(args))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args, argv))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # (dmabuf IPC: the only kind this pool's host driver supports; must be in the environment before HIP starts)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    # the CPU baseline forks a process pool: before torch / HIP are initialised in this process
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    import torch
    import torch.distributed as dist
    ndev = torch.cuda.device_count()          # (does not initialise the GPU)
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if world > ndev and not args.share_gpus:
        raise SystemExit("bench.py: --gpus %d but this node shows %d GPU(s); one rank per GPU "
                         "(--share-gpus wraps ranks onto the cards for a rehearsal)" % (world, ndev))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    local = local % ndev                      # (only differs under --share-gpus)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)      # RCCL over xGMI
        else:
            dist.init_process_group(args.dist_backend)          # gloo: CPU rehearsal of the N > 1 path

    rank_body(args, rank, world, local, cpu, torch, dist, dev, ndev)
    if world > 1:
        dist.destroy_process_group()


def threaded_main(args):
    """`--gpus N --dist-backend threads --share-gpus`: the N ranks as threads of THIS process on the one card
    (vlite-fast_amd/threaded_ranks.py) -- the rehearsal of the N > 1 path (sharding, the coadd leg's indexing over N
    slices, the configs3 sub-record at N = 8) for boxes that allow fewer processes on a GPU than N.  Times mean nothing."""
    if not args.share_gpus:
        raise SystemExit("bench.py: --dist-backend threads puts every rank on one card: a rehearsal, say --share-gpus")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    ndev = torch.cuda.device_count()
    if ndev < 1 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    tr = importlib.import_module("vlite-fast_amd.threaded_ranks")
    dev = torch.device("cuda", 0)

    def body(rank, world, dist):
        torch.cuda.set_device(0)
        rank_body(args, rank, world, 0, None, torch, dist, dev, ndev)
        return 0

    tr.run_as_threads(args.gpus, body)
    return 0


def rank_body(args, rank, world, local, cpu, torch, dist, dev, ndev):
    """what one rank measures and (rank 0) prints, inside an initialised process group when world > 1"""
    lp = importlib.import_module("vlite-fast_amd.libpb")
    S, A = args.seg_per_step, args.ant_per_gpu
    if args.coadd_selftest and world == 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 200))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        plain = run_chain(torch, dist, lp, args, dev, local, 0, 1, args.taps, args.steps, args.warmup, coadd=False)
        withc = run_chain(torch, dist, lp, args, dev, local, 0, 1, args.taps, args.steps, args.warmup, coadd=True)
        print(json.dumps({"coadd_selftest": {"ms_per_step_plain": round(plain["ms_per_step"], 4),
                                             "ms_per_step_with_coadd_leg": round(withc["ms_per_step"], 4),
                                             "emulated_world": args.emulate_world or 1, "antennas_per_gpu": A,
                                             "emulated_layout": ("sliced" if args.coadd_layout != "root" else "root") if args.emulate_world > 1 else None,
                                             "stage_ms_per_step": withc["stage_ms_per_step"]}}))
        dist.destroy_process_group()
        return
    r = run_chain(torch, dist, lp, args, dev, local, rank, world, args.taps, args.steps, args.warmup)
    # BASELINE configs[3] on the whole node: 16 antennas = 2 per GPU on 8 GPUs (every rank takes part)
    c3 = None
    c3_at = int(os.environ.get("PB_BENCH_CONFIGS3_AT", "8"))      # (rehearsals: the world size that adds the sub-record)
    if world == c3_at and world > 1 and A == 1 and not args.no_extras and args.taps == 1:
        c3 = run_chain(torch, dist, lp, args, dev, local, rank, world, 1, args.steps, args.warmup, ant_per_gpu=2)

    if rank == 0:
        msamp, nant_total = r["msamp"], r["nant_total"]
        which = ("configs[1]" if world == 1 and A == 1 else
                 "configs[3] (16 antennas over 8 GPUs)" if world * A == 16 and world == 8 else
                 "configs[3] sharding, %d antenna(s) per GPU" % A if world > 1 else "configs[2]-style batch")
        out = {
            "metric": "Msamp/s/antenna (dual-pol) and x real-time @128 MS/s; % HBM roofline",
            "value": round(msamp, 1), "unit": "Msamp/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "precondition_steps": PRECONDITION_STEPS, "ms_per_step": round(r["ms_per_step"], 4),
            "ms_per_step_cold": round(r["ms_per_step_cold"], 4), "timed_regions": r["regions"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic genbase-style 8-bit Gaussian baseband (mean 128.5, sigma 16.9 codes), "
                    "%g%% of 500-sample blocks with impulsive RFI; resident in HBM" % (100 * args.rfi_frac),
            "config": {"workload": "%s: %d antenna/GPU, 128 MS/s dual-pol, 1 s per step "
                                   "(10 x 100-ms segments, 2048 x 12500-pt FFT rows each), RFI mode %d, "
                                   "%d-bit out, taps=%d, %s FFT" % (which, A, args.rfi_mode, args.nbit, args.taps, args.backend),
                       "antennas": nant_total, "antennas_per_gpu": A, "segments_per_step": S,
                       "parallelism": "antenna-per-GPU" + (("+rccl-gather-tree-coadd" if args.coadd_order == "tree"
                                                                        else "+rccl-reduce-coadd") if world > 1 else "")},
            "msamp_per_antenna": round(msamp / nant_total, 1),
            "x_realtime_per_antenna": round(msamp / nant_total / 128.0, 1),
            "x_realtime_per_antenna_cold": round(msamp / nant_total / 128.0 * r["ms_per_step"] / r["ms_per_step_cold"], 1),
            "roofline": r["roofline"],
            "stage_ms_per_step": r["stage_ms_per_step"],
        }
        if world > 1:
            out["rccl_ranks"] = dist.get_world_size()
            out["dist_backend"] = "%s (%s)" % (dist.get_backend(), "RCCL" if args.dist_backend == "nccl" else "rehearsal")
            out["gpus_visible"] = ndev
            out["per_rank"] = r.get("per_rank")
            out["reduce_leg"] = r.get("reduce_leg")
            out["coadd_order"] = {"order": args.coadd_order, "layout": r.get("coadd_layout"),
                                  "meaning": ("antennas split by index parity, recursively (DESIGN.md section 6): every rank's "
                                              "node by pb_coadd_local_tree; " +
                                              ("an all-to-all of plane slices, every rank evaluates the top levels "
                                               "(pb_coadd_tree) and requantises (pb_coadd_digitise) 1/N of the plane, code "
                                               "bytes gathered to rank 0" if r.get("coadd_layout") == "sliced" else
                                               "one fp32 plane per rank gathered to rank 0, pb_coadd_tree + requantisation "
                                               "there") + "; bytes independent of the number of GPUs")
                                  if args.coadd_order == "tree" else
                                  "left-to-right local sums, one RCCL reduce(SUM): association left to the collective"}
            # one rank per GPU, all of them in the group: anything else is not the run the line claims to be
            if not args.share_gpus:
                assert out["rccl_ranks"] == out["n_gpus"] == args.gpus and ndev >= world, (out["rccl_ranks"], out["n_gpus"], ndev)
            out["ranks_ok"] = bool(out["rccl_ranks"] == out["n_gpus"] and ndev >= world)
        if c3 is not None:
            out["configs3"] = {"antennas": c3["nant_total"], "antennas_per_gpu": 2, "value": round(c3["msamp"], 1),
                               "unit": "Msamp/s", "ms_per_step": round(c3["ms_per_step"], 4),
                               "ms_per_step_cold": round(c3["ms_per_step_cold"], 4), "timed_regions": c3["regions"],
                               "x_realtime_per_antenna": round(c3["msamp"] / c3["nant_total"] / 128.0, 1),
                               "stage_ms_per_step": c3["stage_ms_per_step"],
                               "per_rank": c3.get("per_rank"), "reduce_leg": c3.get("reduce_leg"),
                               "product": "vlite-fast_amd/coadd_host.py runs this leg (coadd.IncoherentCoadd) on antenna "
                                          "dumps / rings and writes the one station-99 .fil",
                               "note": "BASELINE configs[3] (16 antennas on 8 GPUs): %d antennas here, 2 per GPU, the fp32 "
                                       "planes summed in the order \"%s\" (%s); rank 0 hands out the coadded "
                                       "second" % (c3["nant_total"], args.coadd_order,
                                                           ("local tree, all-to-all of slices, every rank sums and requantises "
                                                            "its share, codes gathered" if c3.get("coadd_layout") == "sliced"
                                                            else "local tree, gather, root tree") if args.coadd_order == "tree"
                                                           else "local sum, one reduce")}
        if world == 1:
            try:
                out["roofline"]["alone"] = alone_record(torch, lp, args, dev, local, args.taps)
            except Exception as e:
                out["roofline"]["alone"] = {"error": str(e)}
            if not args.no_power:
                try:
                    out["roofline"]["power"] = power_record(torch, lp, args, dev, local, args.taps)
                except Exception as e:
                    out["roofline"]["power"] = {"error": str(e)}
            out["roofline"]["residency"] = None if args.no_residency else residency_record()
            try:
                pw = out["roofline"].get("power") or {}
                out["roofline"]["valu"] = valu_record(args, args.taps, r["ms_per_step"], pw.get("gfx_mhz"))
            except Exception as e:
                out["roofline"]["valu"] = {"error": str(e)}
        if world == 1 and not args.no_extras and args.taps == 1 and A == 1:
            nsub = max(20, 2 * args.steps // 3)      # (short runs time the pipeline's fill and drain)
            t4 = run_chain(torch, dist, lp, args, dev, local, rank, world, 4, nsub, min(args.warmup, 5))
            out["taps4"] = {"ms_per_step": round(t4["ms_per_step"], 4), "ms_per_step_cold": round(t4["ms_per_step_cold"], 4),
                            "timed_regions": t4["regions"], "value": round(t4["msamp"], 1), "unit": "Msamp/s",
                            "x_realtime_per_antenna": round(t4["msamp"] / 128.0, 1),
                            "x_realtime_per_antenna_cold": round(t4["msamp"] / 128.0 * t4["ms_per_step"] / t4["ms_per_step_cold"], 1),
                            "steps": nsub,
                            "roofline": dict(t4["roofline"], alone=alone_record(torch, lp, args, dev, local, 4)),
                            "stage_ms_per_step": t4["stage_ms_per_step"],
                            "note": "4-tap Hamming WOLA window (analysis/baseband.py:1207-1237) in the streaming path"}
            if cpu is not None and cpu.get("pfb") is not None:
                out["taps4"]["cpu_baseline"] = cpu.pop("pfb")       # P2 timed beside the mode it is the baseline of
            ing = run_chain(torch, dist, lp, args, dev, local, rank, world, 1, nsub, min(args.warmup, 5), ingest=True, regions=1)
            gbs = 2 * (S * ing_frames(S) * 5032) / (ing["ms_per_step"] * 1e-3) / 1e9
            out["ingest"] = {"ms_per_step": round(ing["ms_per_step"], 4), "value": round(ing["msamp"], 1), "unit": "Msamp/s",
                             "x_realtime_per_antenna": round(ing["msamp"] / 128.0, 1), "steps": nsub,
                             "h2d_GBps": round(gbs, 1), "stage_ms_per_step": ing["stage_ms_per_step"],
                             "note": "each second = 51 200 VDIF frames (257.6 MB) from page-locked host memory through "
                                     "pb_submit_vdif (H2D + in-kernel deframe), pipelined over 2 buffer sets: "
                                     "PCIe-bound; not the headline value"}
            try:
                out["search"] = search_record(lp, torch)
            except Exception as e:      # the search stage is not part of the headline path
                out["search"] = {"error": str(e)}
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))


def ing_frames(S):
    """VDIF frames per thread per segment"""
    return ROWS * NFFT // 5000


if __name__ == "__main__":
    main()
