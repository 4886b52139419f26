"""Process-wide libdriftmi context (one GPU per process)."""
import os

_ctx = None


def get_context(workspace_bytes=None):
    """The Context bound to this process's GPU (LOCAL_RANK selects it).  Raises if no GPU."""
    global _ctx
    if _ctx is None:
        from ._lib import Context

        dev = int(os.environ.get("LOCAL_RANK", "0"))
        if workspace_bytes is None:
            workspace_bytes = int(float(os.environ.get("DRIFTMI_WORKSPACE_GB", "8")) * (1 << 30))
        _ctx = Context(dev, workspace_bytes=workspace_bytes)
    return _ctx


def reset_context():
    global _ctx
    if _ctx is not None:
        _ctx.close()
    _ctx = None
