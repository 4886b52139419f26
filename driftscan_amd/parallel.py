"""Process-level parallelism over independent m-blocks.

The reference distributes m over MPI ranks through ``caput.mpiutil``
(drift/core/beamtransfer.py:720, kltransform.py:496, :21-46).  Here one process
drives one GPU and the ranks talk through ``torch.distributed`` — backend "nccl"
(= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.  m-blocks need no data-path
collective at all: only small spectra are gathered to rank 0 and the Fisher
matrix is all-reduced, exactly the reference's pattern.
"""
import os
import time

import numpy as np

# wall seconds this process spent inside collectives (barrier, gathers, broadcast, all-reduce) and how many it made:
# what `bench.py`'s N-rank north-star job reports as `collective_s` (waiting for the slowest rank is part of it)
_coll = dict(seconds=0.0, calls=0, allreduce_s=0.0, allreduce_calls=0)


def collective_stats(reset=False):
    out = dict(_coll)
    if reset:
        for k in _coll:
            _coll[k] = 0 if k.endswith("calls") else 0.0
    return out


class _timed(object):
    def __init__(self, kind=None):
        self.kind = kind

    def __enter__(self):
        self.t0 = time.perf_counter()

    def __exit__(self, *exc):
        dt = time.perf_counter() - self.t0
        _coll["seconds"] += dt
        _coll["calls"] += 1
        if self.kind == "allreduce":
            _coll["allreduce_s"] += dt
            _coll["allreduce_calls"] += 1
        return False


_objgrp = None


def _obj_group(d):
    """Process group for the PICKLED-object collectives (spectra gathers, broadcasts): the default group on gloo; with
    the default group on "nccl" a gloo side group made at first use — the reference gathers pickles over MPI on the
    host (kltransform.py:29), RCCL is for the tensor collective of the path (the Fisher all-reduce).  Every rank
    reaches its first object collective at the same point of the job, so the group creation is collective too."""
    global _objgrp
    if d.get_backend() != "nccl":
        return None
    if _objgrp is None or _objgrp[0] is not d.group.WORLD:
        _objgrp = (d.group.WORLD, d.new_group(backend="gloo"))
    return _objgrp[1]


def _dist():
    import torch.distributed as dist

    return dist if (dist.is_available() and dist.is_initialized()) else None


# One rank's SHARE of an N-rank job in a single process (no process group): partitions are those of rank r of N, the
# gathers and reductions see this rank's part only (rank 0 writes the collected files from what it has).  m-blocks
# are independent and the path has no data-path collective, so the share's wall time is the N-rank job's —
# `bench.py --workload configs2 --share r/N` measures the north-star job this way on a one-GPU box.
_virtual = None


def set_virtual(r=None, n=None):
    """Emulate rank `r` of `n` (None: off) — for measurements only (`bench.py --share r/n`): while it is on, the
    collected files and the `beam_m/COMPLETED` marker are NOT written (they would describe one share as the whole job)."""
    global _virtual
    _virtual = None if r is None else (int(r), int(n))


def _virt():
    return _virtual


def is_virtual():
    return _virtual is not None and _dist() is None


def rank():
    d = _dist()
    if d:
        return d.get_rank()
    v = _virt()
    return v[0] if v else 0


def size():
    d = _dist()
    if d:
        return d.get_world_size()
    v = _virt()
    return v[1] if v else 1


def rank0():
    return rank() == 0


def io_root():
    """The process that creates the shared directories and the configuration copies: rank 0 of a process group — or
    this process when it emulates a share on its own (`set_virtual`: there is nobody else to do it)."""
    return rank0() or _dist() is None


def _bind_device():
    """nccl collectives run on torch's CURRENT device, which is only set once a driftmi Context exists; barriers are reached
    earlier (ProductManager.from_config, BeamTransfer.__init__, _generate_dirs).  Without this every rank would report
    device 0 there and the first RCCL barrier of a multi-GPU job would fail with 'Duplicate GPU detected'.  The device is
    `device.device_index()` — the same choice the contexts make."""
    import torch

    from . import device

    dev = device.device_index()
    if torch.cuda.current_device() != dev:
        torch.cuda.set_device(dev)
    return dev


def barrier():
    d = _dist()
    if d:
        with _timed():
            if d.get_backend() == "nccl":
                d.barrier(device_ids=[_bind_device()])
            else:
                d.barrier()


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* if a launcher set them."""
    import torch
    import torch.distributed as dist

    if dist.is_initialized() or int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend == "nccl":
        _bind_device()
    dist.init_process_group(backend=backend)


def partition(items, costs=None):
    """The items this rank owns.

    Without costs: the reference's contiguous split (``partition_list_mpi``).  With
    costs: longest-processing-time greedy assignment — the cost of an m-block falls
    steeply with m, so the contiguous split leaves most ranks idle at the end; any
    assignment yields the same products, so this is a pure scheduling choice.
    """
    items = list(items)
    n, r = size(), rank()
    if n == 1:
        return items
    if costs is None:
        base, rem = divmod(len(items), n)
        start = r * base + min(r, rem)
        return items[start : start + base + (1 if r < rem else 0)]
    order = np.argsort(-np.asarray(costs, dtype=np.float64), kind="stable")
    load = np.zeros(n)
    mine = []
    for idx in order:
        tgt = int(np.argmin(load))
        load[tgt] += costs[idx]
        if tgt == r:
            mine.append(items[idx])
    return sorted(mine, key=items.index)


def partition_for(items, r, n):
    """The contiguous equal-count slice rank r of n owns (what `partition(items)` returns on that rank)."""
    items = list(items)
    base, rem = divmod(len(items), n)
    start = r * base + min(r, rem)
    return items[start : start + base + (1 if r < rem else 0)]


def partition_contiguous(items, costs, n=None, r=None):
    """Contiguous ranges of `items` with (nearly) equal summed cost: the slice this rank owns.  Beam-transfer
    generation produces a contiguous range of m per call, so a rank that keeps its blocks in HBM from
    generation to the KL stage owns a range; the boundaries follow the cost model instead of the reference's
    equal-count split (caput.mpiutil.partition_list_mpi).  Every rank gets at least one item while items last."""
    items = list(items)
    n = size() if n is None else n
    r = rank() if r is None else r
    if n == 1:
        return items
    c = np.cumsum(np.asarray(costs, dtype=np.float64))
    total = float(c[-1]) if len(items) else 0.0
    bounds = [0]
    for k in range(1, n):
        b = int(np.searchsorted(c, total * k / n, side="left")) + 1
        b = min(max(b, bounds[-1] + 1), max(len(items) - (n - k), bounds[-1]))
        bounds.append(b)
    bounds.append(len(items))
    return items[bounds[r] : bounds[r + 1]]


def rebalance_contiguous(ranges, times, r=None):
    """One round of measured load balancing of contiguous ranges: `ranges` = [(first, last)] of every rank (a partition of
    0 .. n - 1 in rank order), `times` = what each rank measured for its range.  The cost of an item is taken as its rank's
    time / its rank's item count, and the contiguous partition is recomputed from those costs.  Returns the new (first, last) of
    rank `r` (default: this rank) — or of every rank with r = "all".  Pure function of its arguments: every rank that holds the
    gathered `times` computes the same boundaries (`bench.py --mode sharded` iterates it a few times and keeps the best)."""
    n_items = ranges[-1][1] + 1
    dens = np.zeros(n_items)
    for (a, b), t in zip(ranges, times):
        dens[a : b + 1] = float(t) / (b - a + 1)
    n = len(ranges)
    items = list(range(n_items))
    if r == "all":
        out = []
        for k in range(n):
            mine = partition_contiguous(items, list(dens), n=n, r=k)
            out.append((mine[0], mine[-1]))
        return out
    mine = partition_contiguous(items, list(dens), n=n, r=rank() if r is None else r)
    return (mine[0], mine[-1])


def gather_objects(obj):
    """Gather picklable objects to rank 0 (list over ranks there, None elsewhere)."""
    d = _dist()
    if not d:
        return [obj]
    out = [None] * size() if rank0() else None
    with _timed():
        d.gather_object(obj, out, dst=0, group=_obj_group(d))
    return out


def exchange(parts):
    """Personalised all-to-all of picklable objects: `parts[d]` goes to rank d; returns the list over source ranks of what
    they sent here (the MPI transposes of drift/pipeline/timestream.py:129-185, :700-760 — frequency-partitioned arrays
    regrouped by m and back).  Without a process group (one rank) the single part comes straight back."""
    d = _dist()
    n = size()
    if len(parts) != (n if d else 1) and not (not d and len(parts) == n):
        raise ValueError("exchange: one part per rank expected (%d given, %d ranks)" % (len(parts), n))
    if not d:
        return [parts[rank()]] if len(parts) > 1 else [parts[0]]
    mine = None
    with _timed():
        grp = _obj_group(d)
        for dst in range(n):
            out = [None] * n if rank() == dst else None
            d.gather_object(parts[dst], out, dst=dst, group=grp)
            if rank() == dst:
                mine = out
    return mine


def bcast_object(obj):
    d = _dist()
    if not d:
        return obj
    box = [obj]
    with _timed():
        d.broadcast_object_list(box, src=0, group=_obj_group(d))
    return box[0]


def allreduce_sum(arr):
    """Sum a numpy float64 array over ranks (the Fisher assembly, psestimation.py:506-507)."""
    d = _dist()
    if not d:
        return arr
    import torch

    t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64))
    with _timed("allreduce"):
        if d.get_backend() == "nccl":
            t = t.cuda(_bind_device())
        d.all_reduce(t)
        out = t.cpu().numpy()
    return out
