"""Rehearsal transport "threads": W logical ranks as W threads of ONE process.

Why it exists: BASELINE configs[3] is 8 ranks x 2 antennas.  This pool hands out one-GPU boxes and lets at most six
processes use the card at once, so eight gloo PROCESSES on the one GPU are not allowed there -- but eight THREADS of one
process are, each with its own PbHandle (own streams, own buffer sets) and its own IncoherentCoadd, talking through
torch's in-process threaded process group (torch.testing._internal.distributed.multi_threaded_pg: every
torch.distributed call -- all_to_all_single, gather, all_gather, all_reduce, barrier -- keeps its real signature and
argument checks; the exchange itself is a copy under a lock).  That runs configs[3]'s real shape -- world 8, 16
antennas, the sliced layout's indexing over 8 slices, 2 antennas per handle -- through the HIP path on one card.

It is a REHEARSAL of shapes and ordering, never a measurement: the ranks share one GPU and one interpreter lock.  The
product transport is RCCL (`--dist-backend nccl`), one process per GPU (scripts/start_coadd:20-58 starts one coadder
rank per antenna host the same way).  Like the gloo rehearsal, collectives cross through host memory (coadd.py), so no
device stream of one thread is ever touched by another.  dist.reduce is not offered by the threaded group: the
"fast" order (one reduce) cannot be rehearsed this way, the defined tree order (gather / all-to-all) can.
"""
import sys
import threading


def run_as_threads(world, body, timeout=None):
    """body(rank, world, dist) on `world` threads, each inside an initialised threaded process group (rank = thread
    index).  Returns [body's return value per rank]; the first exception of any rank is re-raised here (the other
    ranks are released by the group's own termination event)."""
    import torch
    import torch.distributed as dist
    from torch.testing._internal.distributed import multi_threaded_pg as tpg

    tpg.ProcessLocalGroup.reset()           # (a termination event left by a run that failed would end this one at once)
    tpg._install_threaded_pg()
    torch._C._distributed_c10d._set_thread_isolation_mode(True)
    store = dist.HashStore()
    results, errors = [None] * world, []

    def worker(rank):
        try:
            dist.init_process_group(backend="threaded", rank=rank, world_size=world, store=store)
            try:
                results[rank] = body(rank, world, dist)
            finally:
                try:
                    dist.destroy_process_group()
                except Exception:
                    pass
        except BaseException as e:      # noqa: B036 -- re-raised in the caller's thread
            errors.append((rank, e, sys.exc_info()[2]))
            try:
                tpg.ProcessLocalGroup.exception_handle(e)      # wake ranks waiting in a collective
            except Exception:
                pass

    threads = [threading.Thread(target=worker, args=(r,), name="rank%d" % r, daemon=True) for r in range(world)]
    try:
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout)
            if t.is_alive():
                raise RuntimeError("threaded rank %s did not finish within %s s" % (t.name, timeout))
    finally:
        torch._C._distributed_c10d._set_thread_isolation_mode(False)
        tpg._uninstall_threaded_pg()
        tpg.ProcessLocalGroup.reset()
    if errors:
        # the rank that failed FIRST and on its own account: the ranks the group's termination event then releases
        # from their collectives leave with a SystemExit of their own
        own = [x for x in errors if not isinstance(x[1], SystemExit)] or errors
        rank, e, tb = own[0]
        raise RuntimeError("threaded rank %d failed: %r" % (rank, e)).with_traceback(tb)
    return results
