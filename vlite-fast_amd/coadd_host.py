#!/usr/bin/env python3
"""Coadder host: BASELINE configs[3] -- N antenna streams sharded over the GPUs of a node, every antenna through the
baseband -> filterbank path, the incoherent sum as ONE fp32 RCCL gather (or reduce) per second to the root GPU, a single coadded
SIGPROC file (station 99) and the coadded ring.

What it replaces in the reference: one `process_baseband ... -C <co_key>` per antenna host writing its excised 8-bit
codes, one write per segment, to the coadd ring (src/process_baseband.cu:1416-1422; ring header with SIGPROC_FILE =
the station-99 name, :272-285, :987), and the external MPI coadder `agdadacoadd` that scripts/start_coadd:16,20-58
starts with one rank per antenna ring under mpirun and that leaves the sum in a ring on the root host.  Here one
process per GPU does both jobs: rank r takes antennas a with a mod world == r (coadd.antennas_of_rank), batches them
in one PbHandle, and per second
    pb_submit_vdif x A -> pb_process -> pb_coadd_local_tree -> dist.gather(fp32, root) -> root: pb_coadd_tree,
    pb_coadd_finish
(coadd.IncoherentCoadd, on a stream of its own, one second behind the batch).  The fp32 additions follow a DEFINED
order (antennas split by index parity, recursively: coadd.py), so the coadded file is the same
bytes whatever the number of ranks; `--coadd-order fast` takes one RCCL reduce instead, whose association is the
collective's.  The root writes
`<datadir>/<stamp>_muos_ea99_kur.fil` (`..._ea99.fil` in RFI mode 0) with the SIGPROC header of
write_sigproc_header (telescope_id 99) and, with -K / --out-sink, the coadded ring; every rank also writes its
antennas' own .fil / _kur.fil exactly as process_baseband does (-w 0 turns those off).
The sum is of the fp32 pre-quantisation planes, scaled by 1/sqrt(N_ant) and requantised once (DESIGN.md section 6:
the reference's coadder consumed already-quantised codes and its arithmetic is not in the repository: unpinned).

Start:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
            vlite-fast_amd/coadd_host.py --replay ant0.uw ant1.uw ... [-b 8 -r 2 -w 2 --datadir D]
   or:  python vlite-fast_amd/coadd_host.py --ranks N --replay ...     (starts those ranks itself, as child processes,
                                                                        before anything touches the GPU)
Antenna inputs: --replay FILE... (one dump per antenna, antenna index = position) or -k KEY... (psrdada ring keys,
hexadecimal like the reference's -k).  Streams are aligned on their VDIF seconds: the sum starts at the latest first
second of all antennas and ends with the first stream that ends (its last second dropped, as everywhere).
"""
import argparse
import importlib
import os
import sys
import time

import numpy as np

_pkg = __package__ or "vlite-fast_amd"
if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
vdif = importlib.import_module(_pkg + ".vdif")
sigproc = importlib.import_module(_pkg + ".sigproc")
dada = importlib.import_module(_pkg + ".dada")
pbmod = importlib.import_module(_pkg + ".process_baseband")

SEG_PER_SEC = pbmod.SEG_PER_SEC
COADD_STATION = 99                                    # get_cofbfile, src/process_baseband.cu:272-285


def build_parser():
    p = argparse.ArgumentParser(prog="coadd_host", description=__doc__.split("\n\n")[0])
    p.add_argument("--replay", nargs="+", default=None, help="one dump file per antenna (antenna = position)")
    p.add_argument("-k", dest="keys_in", nargs="+", type=lambda s: int(s, 16), default=None,
                   help="one psrdada ring key (hex) per antenna")
    p.add_argument("-K", dest="key_out", type=lambda s: int(s, 16), default=0, help="coadded ring (root)")
    p.add_argument("--out-sink", default=None, help="file standing in for ring -K")
    p.add_argument("-o", dest="stdout_output", action="store_true")
    p.add_argument("-w", dest="write_fb", type=int, default=2, help="0: no per-antenna files (the coadded file is always written)")
    p.add_argument("-b", dest="nbit", type=int, default=2)
    p.add_argument("-P", dest="npol", type=int, default=1)
    p.add_argument("-r", dest="rfi_mode", type=int, default=2)
    p.add_argument("--datadir", default=sigproc.DATADIR)
    p.add_argument("--logdir", default=pbmod.LOGDIR)
    p.add_argument("--fft-backend", choices=["lds", "hipfft"], default="lds")
    p.add_argument("--taps", type=int, default=1)
    p.add_argument("--rows-per-seg", type=int, default=1024, help="test hook: shorter segments")
    p.add_argument("--nsets", type=int, default=2)
    p.add_argument("--ranks", type=int, default=0, help="start this many ranks (one per GPU) as child processes")
    p.add_argument("--master-port", type=int, default=0)
    p.add_argument("--dist-backend", default="nccl",
                   help="nccl (= RCCL, default); rehearsals: gloo (processes, collectives through host memory) or threads "
                        "(with --ranks N: the N ranks as threads of THIS process, for boxes that allow few processes on a card)")
    p.add_argument("--coadd-input", choices=["planes", "codes"], default="planes",
                   help="what is summed: the fp32 planes before quantisation (default) or the antennas' quantised codes")
    p.add_argument("--coadd-order", choices=["tree", "fast"], default="tree",
                   help="tree (default): the defined order of the fp32 additions, the same bytes on any number of ranks; "
                        "fast: one reduce(SUM), order left to the collective")
    p.add_argument("--coadd-layout", choices=["auto", "root", "sliced"], default="auto",
                   help="tree order over a power-of-two world: sliced (auto, with -P 1) = every rank sums and requantises "
                        "1/W of the plane, code bytes gathered; root = every plane gathered to rank 0.  The same bytes.")
    p.add_argument("--share-gpus", action="store_true", help="rehearsal only: ranks may wrap onto the cards")
    return p


def launch_ranks(args, argv, run=None):
    """`coadd_host.py --ranks N` with no WORLD_SIZE in the environment: the N ranks as CHILD processes under
    torch.distributed.run (this process never touches the GPU and nothing is exec'd); -> their exit code."""
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
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return (run or subprocess.run)(cmd, env=env).returncode


class Control(object):
    """The few integers the ranks must agree on per second (all still have data?  the same second?), over a
    host-side group so that no rank waits on its GPU for them.  Single process: the identity."""

    def __init__(self, dist, backend):
        self.dist = dist
        self.group = None
        if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
            self.group = dist.new_group(backend="gloo") if backend == "nccl" else dist.group.WORLD
            self.on = True
        else:
            self.on = False

    def _red(self, vals, op):
        if not self.on:
            return [int(v) for v in vals]
        import torch
        t = torch.tensor([int(v) for v in vals], dtype=torch.int64)
        self.dist.all_reduce(t, op=op, group=self.group)
        return [int(v) for v in t.tolist()]

    def min(self, *vals):
        return self._red(vals, self.dist.ReduceOp.MIN if self.on else None)

    def max(self, *vals):
        return self._red(vals, self.dist.ReduceOp.MAX if self.on else None)


def run(args, rank=0, world=1, local=0, rings=None, handle=None, dist=None, device=None, coadd=None):
    """One rank of the coadder.  rings: {antenna: ReadRing} of this rank's antennas (tests); handle / coadd: stand-ins
    (tests without a GPU).  Returns the exit status (0, or 1 after a > 1 s data skip or misaligned streams)."""
    pbmod.validate(argparse.Namespace(nbit=args.nbit, rfi_mode=args.rfi_mode, npol=args.npol, rows_per_seg=args.rows_per_seg,
                                      gpu_id=local))
    cmod = importlib.import_module(_pkg + ".coadd")
    nant = len(args.replay) if args.replay else len(args.keys_in or ())
    if nant < world:
        raise SystemExit("coadd_host: %d antenna stream(s) for %d ranks; every rank needs at least one" % (nant, world))
    mine = cmod.antennas_of_rank(nant, rank, world)
    _log = pbmod.Log(args.logdir, args.stdout_output, suffix="_rank%d" % rank if args.dist_backend == "threads" else "")

    def log(level, msg):
        _log(level, msg)
        if level == "ERR" and not args.stdout_output:       # (a rank that gives up says why where the launcher shows it)
            print("coadd_host rank %d: %s" % (rank, msg), file=sys.stderr)
    log("INFO", "[COADD_HOST] rank %d of %d, antennas %s, invoked with: \n%s" % (rank, world, mine, " ".join(sys.argv)))
    R = args.rows_per_seg
    frames_per_sec = R * SEG_PER_SEC * 12500 // vdif.VD_DAT
    sec_bytes = 2 * frames_per_sec * vdif.VD_FRM
    if rings is None:
        rings = {a: (dada.FileRing([args.replay[a]]) if args.replay else dada.open_ring(args.keys_in[a])) for a in mine}
    A = len(mine)
    own_handle = handle is None
    if handle is None:
        lp = importlib.import_module(_pkg + ".libpb")
        handle = lp.PbHandle(device=local, nant=A, nbit=args.nbit, npol=args.npol, rfi_mode=args.rfi_mode, taps=args.taps,
                             fft_backend=lp.FFT_LDS if args.fft_backend == "lds" else lp.FFT_HIPFFT, rows_per_seg=R,
                             max_seg=SEG_PER_SEC, keep_ave=True, nsets=args.nsets)
    if coadd is None:
        coadd = cmod.IncoherentCoadd(handle, nant, device, root=0, backend=args.dist_backend, source=args.coadd_input,
                                     order=args.coadd_order, layout=args.coadd_layout)
    log("INFO", "Incoherent sum: %s of %d antennas over %d rank(s), order \"%s\", layout \"%s\"."
        % (args.coadd_input, nant, world, getattr(coadd, "order", args.coadd_order), getattr(coadd, "layout", "root")))
    ctl = Control(dist, args.dist_backend)
    trim, nsets = handle.trim, handle.nsets
    root = rank == 0
    out_ring = None
    if root and (args.out_sink or args.key_out):
        out_ring = dada.FileSink(args.out_sink) if args.out_sink else dada.open_ring(args.key_out, "w")
    blocks = None
    exit_status = 0
    written = []
    tsamp = 12500.0 / 128e6 * 8

    open_files = []
    try:
        while True:
            # ---- observation set-up: headers, first frames, the common start second
            log("INFO", "Waiting for DADA headers.")
            hdrs, firsts = {}, {}
            ok = 1
            for a in mine:
                raw_hdr = rings[a].next_header()
                if raw_hdr is None:
                    ok = 0
                    break
                hdrs[a] = vdif.ascii_header_parse(raw_hdr)
                f = rings[a].read(vdif.VD_FRM)
                if len(f) != vdif.VD_FRM:
                    log("ERR", "Problem reading first bloody frame!  Bailing.")
                    ok = 0
                    break
                firsts[a] = f
            if not ctl.min(ok)[0]:
                log("INFO", "An input ring closed.  Exiting.")
                break
            log("INFO", "Beginning new observation.")
            vhs = {a: vdif.unpack_header(firsts[a]) for a in mine}
            e_lo, = ctl.min(min(v["epoch"] for v in vhs.values()))
            e_hi, s0 = ctl.max(max(v["epoch"] for v in vhs.values()), max(v["second"] for v in vhs.values()))
            if e_lo != e_hi:
                log("ERR", "Antenna streams carry different VDIF epochs (%d, %d)!" % (e_lo, e_hi))
                exit_status = 1
                break
            if blocks is None:
                blocks = [[pbmod._pinned_bytes(sec_bytes) for _ in range(nsets + 1)] for _ in mine]
            readers = [pbmod.SecondReader(rings[a], firsts[a], sec_bytes, log) for a in mine]
            for i in range(A):
                handle.reset_history(i)
            # antennas that started early skip whole seconds up to the common start
            ok = 1
            for i, rd in enumerate(readers):
                while ok and rd.current_sec < s0:
                    have, nh = rd.fill(blocks[i][0])
                    if have is None:
                        ok = 0
                    else:
                        rd.advance()
                if rd.current_sec != s0:
                    ok = 0
            if not ctl.min(ok)[0]:
                log("ERR", "Antenna streams do not overlap in time; nothing to coadd.")
                exit_status = 1
                break
            vh0 = dict(vhs[mine[0]], second=s0, frame=0, thread=0)
            t_unix = vdif.vdif_to_unixepoch(vh0)
            dmjd = vdif.frame_dmjd(vh0, frames_per_sec)
            # ---- files: every antenna's own (as process_baseband), the coadded one on the root
            fps = []
            # every antenna's own files are named by (second, STATIONID) alone: two streams with one id would open
            # the same files and interleave their writes
            st_vec = [-1] * nant
            for a in mine:
                st_vec[a] = int(hdrs[a].get("STATIONID", 0))
            st_all = ctl.max(*st_vec)
            if args.write_fb != 0 and len(set(st_all)) != nant:
                dup = sorted(set(x for x in st_all if st_all.count(x) > 1))
                log("ERR", "Antenna streams share STATIONID %s: their filterbank files would collide.  Give the streams "
                           "distinct ids or run with -w 0 (coadded file only)." % dup)
                exit_status = 1
                break
            allow = pbmod.load_write_allow(log=log) if args.write_fb == 1 else None
            for a in mine:
                station = int(hdrs[a].get("STATIONID", 0))
                fb, fb_kur, cofb, cofb_kur = sigproc.fb_names(t_unix, station, args.datadir)
                to_null = args.write_fb == 0
                if args.write_fb == 1:           # the reference's source selection, per antenna (src/process_baseband.cu:880-923)
                    if pbmod.source_allowed(hdrs[a], allow):
                        log("INFO", "Source %s matches target list, recording filterbank data." % hdrs[a].get("NAME", ""))
                    else:
                        to_null = True
                        log("INFO", "Source %s not on target list, disabling filterbank data." % hdrs[a].get("NAME", ""))
                if to_null:
                    fb = fb_kur = os.devnull
                sp = sigproc.sigproc_header(station, float(hdrs[a].get("RA", 0)), float(hdrs[a].get("DEC", 0)),
                                            hdrs[a].get("NAME", ""), dmjd, args.npol, args.nbit)
                main_fp = open(fb if args.rfi_mode in (0, 2) else fb_kur, "wb")
                kur_fp = open(fb_kur, "wb") if args.rfi_mode == 2 else None
                main_fp.write(sp)
                if kur_fp:
                    kur_fp.write(sp)
                fps.append((main_fp, kur_fp))
                open_files.extend(f for f in (main_fp, kur_fp) if f)
                written.append((fb, fb_kur))
            co_fp, co_name = None, None
            if root:
                h0 = hdrs[mine[0]]
                co_name = cofb_kur if args.rfi_mode else cofb
                co_fp = open(co_name, "wb")
                open_files.append(co_fp)
                co_fp.write(sigproc.sigproc_header(COADD_STATION, float(h0.get("RA", 0)), float(h0.get("DEC", 0)),
                                                   h0.get("NAME", ""), dmjd, args.npol, args.nbit))
                log("INFO", "Writing the coadded filterbank of %d antennas to %s." % (nant, co_name))
                if out_ring is not None:
                    oh = dict(h0, STATIONID=str(COADD_STATION))
                    out_ring.write_header(vdif.ascii_header_format(sigproc.psrdada_out_header(
                        oh, vh0, args.npol, args.nbit, co_name, t_unix, vdif.frame_mjd(vh0), vdif.frame_mjd_sec(vh0))))
            st = dict(done=0, co_written=0, fb_bytes=0)
            t_obs = time.time()

            def write_coadded(age):
                v = coadd.coadded(SEG_PER_SEC, age=age)
                if v is None:
                    return
                b = np.array(v, copy=True)
                co_fp.write(b.tobytes())
                if out_ring is not None:
                    out_ring.write(b)
                st["co_written"] += 1

            def collect(k):
                """second k: every local antenna's bytes -> its files; its incoherent-sum leg is queued now (the batch is
                known to be complete), and the root writes the coadded bytes of the second before"""
                handle.select_set(k % nsets)
                for i in range(A):
                    out = handle.fetch(i, 0, SEG_PER_SEC, raw=args.rfi_mode != 1, kur=args.rfi_mode != 0)
                    main_fp, kur_fp = fps[i]
                    main_fp.write((out["raw"] if args.rfi_mode != 1 else out["kur"]).tobytes())
                    if kur_fp:
                        kur_fp.write(out["kur"].tobytes())
                st["fb_bytes"] += SEG_PER_SEC * trim
                coadd.queue(k % nsets, SEG_PER_SEC)
                if root and k >= 1:
                    write_coadded(1)
                st["done"] += 1

            queued = 0
            while True:
                ok, skip = 1, 0
                haves = []
                for i, rd in enumerate(readers):
                    have, nh = rd.fill(blocks[i][queued % (nsets + 1)])
                    if have is None:
                        ok = 0
                        break
                    if nh["second"] - rd.current_sec > 1:
                        log("ERR", "Major data skip!  (%d vs. %d; thread = %d) Aborting this observation."
                            % (nh["second"], rd.current_sec, nh["thread"]))
                        skip = 1
                    if rd.current_sec != s0 + queued:
                        skip = 1
                    haves.append(have)
                ok, nskip = ctl.min(ok, -skip)
                if nskip:
                    exit_status = 1
                    break
                if not ok:
                    break                                   # a stream has ended: its last second is dropped, the sum ends
                handle.select_set(queued % nsets)
                for i, rd in enumerate(readers):
                    blk = blocks[i][queued % (nsets + 1)]
                    rd.seal(blk, haves[i])
                    handle.submit_vdif(i, 0, blk, second=rd.current_sec, frame0=0)
                handle.process(SEG_PER_SEC, 0)
                for rd in readers:
                    rd.advance()
                queued += 1
                if queued - st["done"] >= nsets:
                    collect(st["done"])
            while st["done"] < queued:
                collect(st["done"])
            if root and queued:
                write_coadded(0)
            for main_fp, kur_fp in fps:
                main_fp.close()
                if kur_fp:
                    kur_fp.close()
            if root:
                co_fp.close()
                if out_ring is not None:
                    out_ring.end_of_data()
                nsamp = st["co_written"] * SEG_PER_SEC * trim * (8 // args.nbit) // 4096
                log("INFO", "Wrote %.2f MB (%.2f s) to %s" % (st["co_written"] * SEG_PER_SEC * trim * 1e-6, nsamp * tsamp, co_name))
            for a in mine:
                if hasattr(rings[a], "finish_observation"):
                    rings[a].finish_observation()
            log("INFO", "Proc Time...%.3f" % (time.time() - t_obs))
            if exit_status:
                break
    finally:
        # (also on an exception: nothing of the leg -- its stream, its buffers -- stays referenced by the handle,
        #  and no file is left open)
        for fp in open_files:
            if not fp.closed:
                fp.close()
        if hasattr(coadd, "close"):
            coadd.close()
        if own_handle:
            handle.close()
    run.last_files = written
    return exit_status


def run_threaded(args):
    """`--dist-backend threads --ranks N`: the N ranks as threads of this process (threaded_ranks.py), every rank with a
    handle and a coadd leg of its own on the one card -- the rehearsal for boxes that allow fewer processes on a GPU
    than configs[3] has ranks.  -> the largest exit status of the ranks."""
    import torch
    if torch.cuda.device_count() < 1 or not torch.cuda.is_available():
        raise SystemExit("coadd_host needs a GPU: the HIP path has no CPU fallback")
    if not args.share_gpus:
        raise SystemExit("coadd_host: --dist-backend threads puts every rank on one card: a rehearsal, say --share-gpus")
    tr = importlib.import_module(_pkg + ".threaded_ranks")
    world = max(1, args.ranks)
    dev = torch.device("cuda", 0)

    def body(rank, world, dist):
        torch.cuda.set_device(0)
        return run(args, rank=rank, world=world, local=0, dist=dist if world > 1 else None, device=dev)

    return max(tr.run_as_threads(world, body))


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = build_parser().parse_args(argv)
    if not args.replay and not args.keys_in:
        raise SystemExit("coadd_host: give the antenna streams with --replay FILE... or -k KEY...")
    if args.dist_backend == "threads":
        sys.exit(run_threaded(args))
    if "WORLD_SIZE" not in os.environ and args.ranks > 1:
        sys.exit(launch_ranks(args, argv))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.ranks and world != args.ranks:
        raise SystemExit("coadd_host: --ranks %d but the launcher started %d rank(s)" % (args.ranks, world))
    import torch
    import torch.distributed as dist
    ndev = torch.cuda.device_count()
    if ndev < 1 or not torch.cuda.is_available():
        raise SystemExit("coadd_host needs a GPU: the HIP path has no CPU fallback")
    if world > ndev and not args.share_gpus:
        raise SystemExit("coadd_host: %d ranks but this node shows %d GPU(s); one rank per GPU "
                         "(--share-gpus wraps ranks onto the cards for a rehearsal)" % (world, ndev))
    local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)
    try:
        rc = run(args, rank=rank, world=world, local=local, dist=dist if world > 1 else None, device=dev)
    finally:
        if world > 1:
            dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
