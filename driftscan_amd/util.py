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
    """Remember the result of the most recent call (same args -> same object back).  The (arguments, result)
    pair is replaced in one assignment, so a reader on another thread (the background file writers call the
    accessors too) can never pair one call's arguments with another call's result."""
    state = [None]  # (args, kwargs, ret) of the last call

    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        last = state[0]
        if last is None or args != last[0] or kwargs != last[1]:
            last = (args, kwargs, func(*args, **kwargs))
            state[0] = last
        return last[2]

    return wrapper
