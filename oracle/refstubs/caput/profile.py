import contextlib


class IOUsage(contextlib.AbstractContextManager):
    def __init__(self, logger=None):
        pass

    def __exit__(self, *a):
        return False


class Profiler(IOUsage):
    pass
