"""libdriftmi contexts: one per (process, thread).

One process drives one GPU (LOCAL_RANK selects it).  A context owns a HIP stream and a workspace
arena; independent groups of m-blocks can be pushed through the library from different Python
threads (ctypes releases the GIL during the calls), each on its own context, so that the
latency-bound phases of one group overlap with the bandwidth- or MFMA-bound phases of another."""
import os
import threading

_local = threading.local()
_all = []
_lock = threading.Lock()
_default_ws = None
_gen = 0  # bumped by reset_context so that every thread drops its closed context


def device_index():
    """The GPU of this process: one process per GPU, LOCAL_RANK picks it; DRIFTMI_DEVICE overrides (several ranks on one
    card in tests).  The ONE place the choice is made: contexts and the nccl collectives of `parallel` both ask here."""
    return int(os.environ.get("DRIFTMI_DEVICE", os.environ.get("LOCAL_RANK", "0")))


def get_context(workspace_bytes=None):
    """The Context of the calling thread (created on first use).  Raises if no GPU."""
    global _default_ws
    ctx = getattr(_local, "ctx", None)
    if ctx is not None and getattr(_local, "gen", -1) != _gen:
        ctx = None
    if ctx is None:
        from ._lib import Context

        dev = device_index()
        if threading.current_thread() is not threading.main_thread() and os.environ.get("DRIFTMI_THREAD_STREAMS", "0") == "1":
            # OPT-IN (DRIFTMI_THREAD_STREAMS=1; `bench.py --streams N` sets it): a worker thread driving its own group of
            # m-blocks gets a HIP stream of its own (torch's pool streams are non-blocking: no implicit join with the default
            # stream), made this thread's current stream so that the library's kernels and torch's copies / allocations of
            # the thread stay in one order.  What the thread takes from another stream it must wait for itself (`wait_for`).
            # Off by default: every context then binds the default stream, and tensors handed between threads keep the
            # implicit ordering callers of the operator API may rely on.
            import torch

            torch.cuda.set_device(dev)
            st = torch.cuda.Stream(device=dev)
            torch.cuda.set_stream(st)
            _local.stream = st
        if workspace_bytes is None:
            workspace_bytes = _default_ws
        if workspace_bytes is None:
            workspace_bytes = int(float(os.environ.get("DRIFTMI_WORKSPACE_GB", "8")) * (1 << 30))
        if _default_ws is None:
            _default_ws = workspace_bytes
        ctx = Context(dev, workspace_bytes=workspace_bytes)
        _local.ctx = ctx
        _local.gen = _gen
        with _lock:
            _all.append(ctx)
    return ctx


def wait_for(event):
    """Make the calling thread's stream wait for a torch.cuda.Event recorded on another stream (no host wait)."""
    import torch

    torch.cuda.current_stream(device_index()).wait_event(event)


def reset_context():
    """Close every context created so far (all threads)."""
    global _default_ws, _gen
    with _lock:
        for c in _all:
            c.close()
        del _all[:]
        _gen += 1
    _local.ctx = None
    _default_ws = None
