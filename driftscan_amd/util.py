"""Filename patterns and a last-call cache (drift/util/util.py:6-32)."""
import functools
import math


def intpattern(n):
    """printf pattern for signed integers up to ``n`` (always shows the sign)."""
    return "%+0" + repr(int(math.ceil(math.log10(n + 1))) + 1) + "d"


def natpattern(n):
    """printf pattern for naturals up to ``n``, zero padded: ceil(log10(n+1)) digits."""
    return "%0" + repr(int(math.ceil(math.log10(n + 1)))) + "d"


def cache_last(func):
    """Remember the result of the most recent call (same args -> same object back)."""
    state = {"args": None, "kwargs": None, "ret": None, "set": False}

    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        if not state["set"] or args != state["args"] or kwargs != state["kwargs"]:
            state["ret"] = func(*args, **kwargs)
            state["args"], state["kwargs"], state["set"] = args, kwargs, True
        return state["ret"]

    return wrapper
